// Single-pass fused backward of the pointsf scorer: dZ chain + every weight gradient in ONE kernel that reads X and the stored
// activations exactly once and never materialises dZ in HBM.
//
// Reference: the autograd backward of ptranking/base/point_ranker.py:45-55 + ptranking/base/utils.py:288-356
// ((Dropout -> Linear -> ReLU) x NL -> Linear) — per hidden layer three ATen kernels and two skinny GEMMs (dX, dW).
//
// HBM traffic per document (F = 136, NL = 3): read 4F (X) + NL*448 (activations) + 4 (dpreds) = 1.9 KB; the only writes are
// one partial gradient per workgroup (256 x 136 KB).  The layer-wise path (scorer.hip) moves 5.8 KB per document.
//
// Structure.  A persistent 8-wave workgroup per CU walks slabs of 32 documents.  Per slab everything lives in LDS as plain
// [row][feature] images (row stride 112 floats = 16 mod 32 banks, X: 16*NT1 (+16)):
//     A_l   the stored activations, landed by LDS-DMA (global_load_lds_dwordx4) one slab ahead, no registers involved
//     XS    the input features with the input dropout recomputed (global -> registers one slab ahead -> hash -> LDS)
//     Z_l   dZ of the hidden layers, produced in place (top layer) or by the chain phases
// and is consumed by fp32 MFMAs in two roles:
//   * dZ chain ("transposed world", as the forward): dA^T[k][row] = W_l^T[k][out] * dZ_l^T[out][row].  Wave w (0..6) owns the
//     16-feature output tile w of every layer and keeps its W_l^T fragments in REGISTERS for the whole kernel (25 k-steps per
//     layer: 24 from six ds_read_b128 + one tail step for features 96..99 — the contraction dimension is not padded to 112);
//     the B operand is a ds_read_b128 of the previous dZ image, the result is gated by the stored activation and written
//     back to LDS as the next dZ image.
//   * weight gradients ("row-contraction world"): dW_l[out][in] = sum_rows dZ_l[row][out] * A_{l-1}[row][in] with rows as
//     the MFMA k index: both operands are ds_read_b32 of the images (lane = feature, lane>>4 = row), i.e. the transpose
//     the row contraction needs is done by the LDS addressing.  The 7*NT1 + 49*(NL-1) accumulator tiles are distributed over
//     the 8 waves at COMPILE time (TileMap): wave 7 owns no chain tile and instead multiplies a 3x2 block of dW_l during the
//     chain phase in which dZ_l is already available, so every wave issues the same number of MFMAs between two barriers.
// NL + 1 workgroup barriers per slab; 994 MFMAs per 16 documents (layer-wise kernels: 1036 + the 7.1x HBM traffic).
// Results are bit-stable: fixed tile ownership, fixed row order, per-workgroup partials reduced by reduce_partials_kernel.
#include "ptr_mlp.h"

namespace ptr {

constexpr int kSR = 32;            // rows per slab
constexpr int kBW = 8;             // waves per workgroup
constexpr int kBT = kBW * 64;
constexpr int kSlabF = kSR * kAL;  // floats of one activation / dZ image

__host__ __device__ constexpr int ldx_of(int NT1) { return (NT1 & 1) ? 16 * NT1 : 16 * NT1 + 16; }   // = 16 (mod 32)
__host__ __device__ constexpr size_t bwd_fused_lds_floats(int NL, int NT1) {
    return (size_t)2 * NL * kSlabF + (size_t)(NL - 1) * kSlabF + (size_t)kSR * ldx_of(NT1) + 2 * kSR + kHP;
}

// Compile-time distribution of the dW accumulator tiles (l, nt, mt) over waves and phases.
//   phase c < CH: the chain phase producing dZ_{NL-2-c}; only wave 7 multiplies dW tiles in it (a 3x2 block of layer NL-1-c)
//   phase CH:     all waves, the remaining tiles in (l, nt, mt) order cut into 8 contiguous runs
template <int NL, int NT1> struct TileMap {
    static constexpr int CH = NL - 1;
    static constexpr int NPH = CH + 1;
    static constexpr int NT = 7 * NT1 + (NL - 1) * 49;
    static constexpr int MAXL = 64;
    int tl[NT], tn[NT], tm[NT], slot[NT];
    int cnt[kBW][NPH];
    int lst[kBW][NPH][MAXL];
    int ntiles[kBW];
    static constexpr int ntl(int l) { return l == 0 ? NT1 : 7; }
    constexpr TileMap() : tl{}, tn{}, tm{}, slot{}, cnt{}, lst{}, ntiles{} {
        int id = 0;
        for (int l = 0; l < NL; ++l)
            for (int n = 0; n < ntl(l); ++n)
                for (int m = 0; m < 7; ++m) { tl[id] = l; tn[id] = n; tm[id] = m; ++id; }
        bool taken[NT] = {};
        for (int c = 0; c < CH; ++c) {
            const int l = NL - 1 - c;
            for (int t = 0; t < NT; ++t)
                if (tl[t] == l && tn[t] >= ntl(l) - 2 && tm[t] < 3) { taken[t] = true; lst[7][c][cnt[7][c]++] = t; }
        }
        int rem = 0;
        for (int t = 0; t < NT; ++t) rem += taken[t] ? 0 : 1;
        const int base = rem / kBW, extra = rem % kBW;
        int w = 0, inw = 0;
        for (int t = 0; t < NT; ++t) {
            if (taken[t]) continue;
            lst[w][CH][cnt[w][CH]++] = t;
            if (++inw == base + (w < extra ? 1 : 0)) { ++w; inw = 0; }
        }
        for (int v = 0; v < kBW; ++v) {
            int s = 0;
            for (int ph = 0; ph < NPH; ++ph)
                for (int k = 0; k < cnt[v][ph]; ++k) slot[lst[v][ph][k]] = s++;
            ntiles[v] = s;
        }
    }
    // first use of the dZ fragment (l, mt) / the A fragment (l, nt) inside a wave's phase list => it has to be read from LDS
    constexpr bool first_a(int w, int ph, int k) const {
        const int t = lst[w][ph][k];
        for (int i = 0; i < k; ++i) { const int u = lst[w][ph][i]; if (tl[u] == tl[t] && tm[u] == tm[t]) return false; }
        return true;
    }
    constexpr bool first_b(int w, int ph, int k) const {
        const int t = lst[w][ph][k];
        for (int i = 0; i < k; ++i) { const int u = lst[w][ph][i]; if (tl[u] == tl[t] && tn[u] == tn[t]) return false; }
        return true;
    }
};
template <int NL, int NT1> inline constexpr TileMap<NL, NT1> kTM{};

__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
// 64 lanes x 16 bytes, global (per-lane address) -> LDS (wave-uniform base + lane*16).  Invisible to hipcc's waitcnt bookkeeping:
// completion is waited for explicitly (s_waitcnt vmcnt(0) before the publishing barrier).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// LDS writes of this wave complete (lgkmcnt) -> workgroup barrier.  No fence semantics on purpose: a __syncthreads() would also
// drain the LDS-DMA and prefetch loads that are meant to stay in flight across the barrier.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Experiment builds only (-DPTR_BWD_TRACE): shader-clock stamps of workgroup 0's first slabs, one row of 16 stamps per (slab, wave),
// written behind the partial gradients in `ws` (which is sized for 2 partials per CU; the fused kernel uses one).
#ifdef PTR_BWD_TRACE
#define BWD_STAMP(i)                                                                                                    \
    do {                                                                                                                \
        if (blockIdx.x == 0 && lane == 0 && nslab_done < 8)                                                             \
            reinterpret_cast<unsigned long long *>(ws + (size_t)gridDim.x * np_stride)[(nslab_done * kBW + W) * 16 + (i)] = clock64(); \
    } while (0)
#else
#define BWD_STAMP(i) do { } while (0)
#endif

template <int NL, int NT1, int W>
__device__ __forceinline__ void bwd_body(const float *__restrict__ X, const float *__restrict__ P, const float *__restrict__ acts,
                                         const float *__restrict__ dpreds, const MlpArgs a, float *__restrict__ ws, size_t np_stride,
                                         float *smem) {
    using TMt = TileMap<NL, NT1>;
    constexpr int CH = NL - 1, LDX = ldx_of(NT1), NTMAX = NT1 > 7 ? NT1 : 7;
    constexpr int NACC = kTM<NL, NT1>.ntiles[W];
    const int F = a.F, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    float *bufA = smem;                                   // [2][NL][kSlabF]  stored activations (top layer: dZ in place)
    float *Zb = bufA + 2 * NL * kSlabF;                   // [CH][kSlabF]     dZ_0 .. dZ_{NL-2}
    float *XS = Zb + CH * kSlabF;                         // [kSR][LDX]       dropped input features
    float *dsb = XS + kSR * LDX;                          // [2][kSR]         dLoss/dscore
    float *wo = dsb + 2 * kSR;                            // [kHP]            w_out
    const uint32_t bufA_addr = lds_byte_addr(bufA);
    const int nslabs = (R + kSR - 1) / kSR;
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    // ---- one-time: w_out to LDS, this wave's W^T fragments to registers
    for (int i = tid; i < kHP; i += kBT) wo[i] = i < kH ? P[off_wout(NL, F) + i] : 0.0f;
    float wf[CH > 0 ? CH : 1][25];
    if constexpr (W < 7) {
        static_for<CH>([&](auto c_) {
            constexpr int c = c_, l = NL - 1 - c;
            const float *Wl = P + off_W(l, F);
            const int k = 16 * W + j;
            const bool kok = k < kH;
            const int kc = kok ? k : 0;
#pragma unroll
            for (int S = 0; S < 6; ++S)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float v = Wl[(size_t)(16 * S + 4 * g + c4) * kH + kc];
                    wf[c][4 * S + c4] = kok ? v : 0.0f;
                }
            const float v = Wl[(size_t)(96 + g) * kH + kc];
            wf[c][24] = kok ? v : 0.0f;
        });
    }

    // ---- loop-invariant slot geometry of the X staging
    constexpr int XTOT = kSR * 4 * NT1, XSL = (XTOT + kBT - 1) / kBT;
    auto x_slot = [&](int u, int &xr, int &xc) -> bool {      // slot u of this thread -> (slab row, float4 column); recomputed at use
        const int idx = tid + kBT * u;
        const bool in = idx < XTOT;
        xr = in ? idx / (4 * NT1) : 0;
        xc = in ? idx - xr * (4 * NT1) : 0;
        return in;
    };
    f32x4 xraw[XSL];
    float dsraw = 0.0f;

    auto issue_loads = [&](int slab, int buf) {
        const int row0 = slab * kSR;
        static_for<(NL * 14 - W + kBW - 1) / kBW>([&](auto q_) {
            constexpr int k = W + kBW * q_, layer = k / 14, ch = k % 14;
            const int idx4 = ch * 64 + lane;
            const int row = idx4 / 28, c4 = idx4 - row * 28;
            const int gr = min(row0 + row, R - 1);
            const float *src = acts + ((size_t)layer * R + gr) * kAL + 4 * c4;
            const uint32_t dst = bufA_addr + (uint32_t)(((buf * NL + layer) * kSlabF + ch * 256) * 4);
            glds16(src, __builtin_amdgcn_readfirstlane(dst));
        });
#pragma unroll
        for (int u = 0; u < XSL; ++u) {
            int xr, xc;
            x_slot(u, xr, xc);
            const int gr = min(row0 + xr, R - 1);
            const int cc = 4 * xc < F ? 4 * xc : 0;
            xraw[u] = *reinterpret_cast<const f32x4 *>(X + (size_t)gr * F + cc);
        }
        dsraw = dpreds[min(row0 + (tid & (kSR - 1)), R - 1)];      // every thread (unconditional load / use, see P0)
    };

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbv[NL][kMT];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int m = 0; m < kMT; ++m) dbv[l][m] = 0.0f;
    f32x4 dwo4 = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.0f;
    const int f4 = tid % 28, r0t = tid / 28;             // top-layer pass: thread -> (feature group, row) for tid < 448

    // dW tiles of phase PH over the slab's 8 k-steps (4 rows each)
    auto dw_phase = [&](auto ph_, const float *cur) {
        constexpr int PH = ph_;
        constexpr int CNT = kTM<NL, NT1>.cnt[W][PH];
        if constexpr (CNT > 0) {
#pragma unroll 2
            for (int ks = 0; ks < kSR / 4; ++ks) {
                float av[NL][kMT], bv[NL][NTMAX];
                static_for<CNT>([&](auto k_) {
                    constexpr int t = kTM<NL, NT1>.lst[W][PH][k_];
                    constexpr int l = kTM<NL, NT1>.tl[t], n = kTM<NL, NT1>.tn[t], m = kTM<NL, NT1>.tm[t], s = kTM<NL, NT1>.slot[t];
                    if constexpr (kTM<NL, NT1>.first_a(W, PH, k_)) {
                        const float *zsrc = l == NL - 1 ? cur + (NL - 1) * kSlabF : Zb + l * kSlabF;
                        av[l][m] = zsrc[(4 * ks + g) * kAL + 16 * m + j];
                        if constexpr (n == 0) dbv[l][m] += av[l][m];
                    }
                    if constexpr (kTM<NL, NT1>.first_b(W, PH, k_)) {
                        if constexpr (l == 0) bv[l][n] = XS[(4 * ks + g) * LDX + 16 * n + j];
                        else bv[l][n] = cur[(l - 1) * kSlabF + (4 * ks + g) * kAL + 16 * n + j];
                    }
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[l][m], bv[l][n], acc[s], 0, 0, 0);
                });
            }
        }
    };

    int slab = blockIdx.x, p = 0;
    int nslab_done = 0;
    (void)nslab_done;
    issue_loads(slab, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const float dsv = slab * kSR + (tid & (kSR - 1)) < R ? dsraw : 0.0f;
        if (tid < kSR) dsb[tid] = dsv;
    }
    wg_barrier();

    for (; slab < nslabs; slab += gridDim.x, p ^= 1) {
        const int row0 = slab * kSR;
        const int next = slab + gridDim.x;
        const bool has_next = next < nslabs;
        float *cur = bufA + p * NL * kSlabF;
        BWD_STAMP(0);

        // ---- P0: X image (input dropout recomputed), top-layer dZ in place, d w_out / d b_out; then the next slab's loads
#pragma unroll
        for (int u = 0; u < XSL; ++u) {
            // the loaded registers are consumed UNCONDITIONALLY (only the store is predicated): a use inside a branch leaves the
            // load "pending" on the skipping path in hipcc's waitcnt model, and the s_waitcnt vmcnt(0) it then places at the
            // next write of those registers lands behind the LDS-DMA issues below — a full HBM round trip per slab
            int xr, xc;
            const bool in = x_slot(u, xr, xc);
            f32x4 v = xraw[u];
            if (a.p_drop > 0.0f) {
                uint32_t w0, w1;
                drop_bits(a.seed_lo, a.seed_hi, 0, min(row0 + xr, R - 1), xc, w0, w1);
                v = drop4(v, w0, w1, thr, inv_keep);
            }
            const float okf = 4 * xc < F ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= okf;
            if (in) *reinterpret_cast<f32x4 *>(XS + xr * LDX + 4 * xc) = v;
        }
        if (tid < 448) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wo + 4 * f4);
#pragma unroll
            for (int h = 0; h < kSR / 16; ++h) {
                const int r = r0t + 16 * h;
                float *pa = cur + (NL - 1) * kSlabF + r * kAL + 4 * f4;
                const f32x4 h4 = *reinterpret_cast<const f32x4 *>(pa);
                const float ds = dsb[p * kSR + r];
                f32x4 d;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d[c] = (ds * w4[c]) * (h4[c] > 0.0f ? 1.0f : 0.0f);
                    dwo4[c] = fmaf(h4[c], ds, dwo4[c]);
                }
                *reinterpret_cast<f32x4 *>(pa) = d;
            }
        } else if (tid < 448 + kSR) {
            dbo += dsb[p * kSR + tid - 448];
        }
        BWD_STAMP(1);
        if (has_next) issue_loads(next, p ^ 1);
        BWD_STAMP(2);
        wg_barrier();
        BWD_STAMP(3);

        // ---- chain phases: dZ_{l-1} = (W_l^T dZ_l) gated by the stored activation; wave 7 multiplies dW_l tiles instead
        static_for<CH>([&](auto c_) {
            constexpr int c = c_, l = NL - 1 - c;
            if constexpr (W < 7) {
                const float *src = c == 0 ? cur + (NL - 1) * kSlabF : Zb + l * kSlabF;
                float *dst = Zb + (l - 1) * kSlabF;
                const float *gate = cur + (l - 1) * kSlabF;
                // one row tile at a time (24 + 1 operand registers); two accumulators over alternating k-steps keep two
                // independent MFMA chains in flight (dependent-accumulator latency 40 cycles vs 32 issue)
#pragma unroll
                for (int rt = 0; rt < kSR / 16; ++rt) {
                    const int row = 16 * rt + j;
                    f32x4 b[6];
#pragma unroll
                    for (int S = 0; S < 6; ++S) b[S] = *reinterpret_cast<const f32x4 *>(src + row * kAL + 16 * S + 4 * g);
                    const float bt = src[row * kAL + 96 + g];
                    const f32x4 gt = *reinterpret_cast<const f32x4 *>(gate + row * kAL + 16 * W + 4 * g);
                    f32x4 ac0 = f32x4{0.f, 0.f, 0.f, 0.f}, ac1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int S = 0; S < 6; ++S) {
                        ac0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][4 * S + 0], b[S][0], ac0, 0, 0, 0);
                        ac1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][4 * S + 1], b[S][1], ac1, 0, 0, 0);
                        ac0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][4 * S + 2], b[S][2], ac0, 0, 0, 0);
                        ac1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][4 * S + 3], b[S][3], ac1, 0, 0, 0);
                    }
                    ac0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c][24], bt, ac0, 0, 0, 0);
                    f32x4 d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = (ac0[e] + ac1[e]) * (gt[e] > 0.0f ? inv_keep : 0.0f);
                    *reinterpret_cast<f32x4 *>(dst + row * kAL + 16 * W + 4 * g) = d;
                }
            } else {
                dw_phase(c_, cur);
            }
            BWD_STAMP(4 + 2 * c);
            wg_barrier();
            BWD_STAMP(5 + 2 * c);
        });

        // ---- all-wave dW phase
        dw_phase(std::integral_constant<int, CH>{}, cur);
        BWD_STAMP(4 + 2 * CH);

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BWD_STAMP(5 + 2 * CH);          // the next slab's LDS-DMA has landed (issued a whole slab ago)
        {
            const float dsv = next * kSR + (tid & (kSR - 1)) < R ? dsraw : 0.0f;
            if (has_next && tid < kSR) dsb[(p ^ 1) * kSR + tid] = dsv;
        }
        wg_barrier();
        BWD_STAMP(6 + 2 * CH);
        ++nslab_done;
    }

    // ---- epilogue: this workgroup's partial gradient in the flat parameter layout
    float *out = ws + (size_t)blockIdx.x * np_stride;
    static_for<TMt::NPH>([&](auto ph_) {
        constexpr int PH = ph_;
        static_for<kTM<NL, NT1>.cnt[W][PH]>([&](auto k_) {
            constexpr int t = kTM<NL, NT1>.lst[W][PH][k_];
            constexpr int l = kTM<NL, NT1>.tl[t], n = kTM<NL, NT1>.tn[t], m = kTM<NL, NT1>.tm[t], s = kTM<NL, NT1>.slot[t];
            const int K = l == 0 ? F : kH;
            const int k = 16 * n + j;
            float *o = out + off_W(l, F);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int of = 16 * m + 4 * g + c;
                if (of < kH && k < K) o[(size_t)of * K + k] = acc[s][c];
            }
            if constexpr (n == 0) {
                float v = dbv[l][m];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int of = 16 * m + j;
                if (g == 0 && of < kH) out[off_b(l, F) + of] = v;
            }
        });
    });
    // d w_out / d b_out: fixed-order sums over the 16 row slots / 32 rows (Zb and XS are free after the loop's last barrier)
    if (tid < 448) *reinterpret_cast<f32x4 *>(Zb + r0t * kAL + 4 * f4) = dwo4;
    else if (tid < 448 + kSR) XS[tid - 448] = dbo;
    wg_barrier();
    if (tid < kH) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += Zb[r * kAL + tid];
        out[off_wout(NL, F) + tid] = s;
    } else if (tid == kH) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < kSR; ++r) s += XS[r];
        out[off_wout(NL, F) + kH] = s;
    }
}

template <int NL, int NT1>
__global__ void __launch_bounds__(kBT)
mlp_bwd_fused_kernel(const float *__restrict__ X, const float *__restrict__ P, const float *__restrict__ acts,
                     const float *__restrict__ dpreds, MlpArgs a, float *__restrict__ ws, size_t np_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: bwd_body<NL, NT1, 0>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 1: bwd_body<NL, NT1, 1>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 2: bwd_body<NL, NT1, 2>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 3: bwd_body<NL, NT1, 3>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 4: bwd_body<NL, NT1, 4>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 5: bwd_body<NL, NT1, 5>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        case 6: bwd_body<NL, NT1, 6>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
        default: bwd_body<NL, NT1, 7>(X, P, acts, dpreds, a, ws, np_stride, smem); break;
    }
}

// PTR_BWD_FUSED=0 selects the layer-wise kernels (A/B measurements, tests); read on every call so a test can flip it
static int bwd_fused_enabled() {
    const char *e = getenv("PTR_BWD_FUSED");
    return e ? (atoi(e) != 0) : 1;
}

bool bwd_fused_supported(int F, int NL, const void *X, const void *acts) {
    if (!bwd_fused_enabled()) return false;
    const int NT1 = (F + 15) / 16;
    return NL == 3 && NT1 == 9 && F % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(acts) & 15) == 0;
}

int bwd_fused_grid(int R) {
    const int nslabs = (R + kSR - 1) / kSR;
    return nslabs < mlp_num_cus() ? nslabs : mlp_num_cus();
}

int launch_bwd_fused(const float *X, const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws,
                     hipStream_t st, const char *who) {
    const int grid = bwd_fused_grid(a.R);
    const size_t NP = n_params(a.NL, a.F);
    auto go = [&](auto kern, int NT1) -> int {
        const size_t lds = bwd_fused_lds_floats(a.NL, NT1) * sizeof(float);
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBT), lds, st, X, params, acts, dpreds, a, ws, NP);
        return check_hip(hipGetLastError(), who);
    };
    return go(mlp_bwd_fused_kernel<3, 9>, 9);
}

}  // namespace ptr
