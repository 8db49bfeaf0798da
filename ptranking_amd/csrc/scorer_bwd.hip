// Single-pass fused backward of the pointsf scorer: dZ chain + every weight gradient in ONE kernel that reads X and the stored
// activations exactly once and never materialises dZ in HBM.
//
// Reference: the autograd backward of ptranking/base/point_ranker.py:45-55 + ptranking/base/utils.py:288-356
// ((Dropout -> Linear -> ReLU) x NL -> Linear) — per hidden layer three ATen kernels and two skinny GEMMs (dX, dW).
//
// HBM traffic per document (F = 136, NL = 3): read 4F (X) + NL*448 (activations) + 4 (dpreds) = 1.9 KB; the only writes are
// one partial gradient per workgroup (256 x 136 KB).  The layer-wise path (scorer.hip) moves 5.8 KB per document.
//
// Structure.  A persistent 8-wave workgroup per CU walks slabs of 32 documents.  Per slab everything lives in LDS: the activation / dZ
// images TILE-MAJOR like the stored activations (ptr_mlp.h act_off: [row tile of 16][feature tile of 16][row][feature], so a slab image is a
// straight 14 KB copy of global memory and lane (feature j, row g) of an MFMA operand read still hits 32 distinct banks), the X image as
// plain [row][feature] (row stride 16*NT1 (+16) = 16 mod 32 banks):
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
// Bias gradients cost nothing: every A image carries a column of ones behind its last real feature (X: written with the image;
// activations: stored by the forward at feature 100), so column K of a dW tile row IS db.
// NL + 1 workgroup barriers per slab; 994 MFMAs per 16 documents (layer-wise kernels: 1036 + the 7.1x HBM traffic).
// Results are bit-stable: fixed tile ownership, fixed row order, per-workgroup partials reduced by reduce_partials_kernel.
#include "ptr_mlp.h"

namespace ptr {

constexpr int kSR = 32;            // rows per slab
#ifndef BWD_WAVES
#define BWD_WAVES 8
#endif
// waves per workgroup: 8 (2 per SIMD, <= 256 VGPRs; the kernel uses 245).  -DBWD_WAVES=12 builds the three-waves-per-SIMD form (waves 8..11
// multiply dW tiles only, <= 168 VGPRs) — measured r3: the chain waves (50 W^T fragment registers + chain operands + accumulators) do not
// fit, 140 registers spill; an experiment switch, not a product path.
constexpr int kBW = BWD_WAVES;
constexpr int kBT = kBW * 64;
constexpr int kSlabF = kSR * kAL;  // floats of one activation / dZ image
// The short VALU / LDS / VMEM bursts that prepare the next slab run at raised wave priority: beside a partner wave that streams
// MFMAs they otherwise only get the issue slots the stream leaves over (measured 4x slower than alone).
#ifndef BWD_PREP_PRIO
#define BWD_PREP_PRIO 1
#endif
#if BWD_PREP_PRIO
#define PREP_BEGIN() __builtin_amdgcn_s_setprio(2)
#define PREP_END() __builtin_amdgcn_s_setprio(0)
#else
#define PREP_BEGIN() do { } while (0)
#define PREP_END() do { } while (0)
#endif
#ifndef BWD_KS_UNROLL
#define BWD_KS_UNROLL 2            // unroll of the dW k-step loop (register pressure vs load batching)
#endif

__host__ __device__ constexpr int ldx_of(int NT1) { return (NT1 & 1) ? 16 * NT1 : 16 * NT1 + 16; }   // = 16 (mod 32)
__host__ __device__ constexpr size_t bwd_fused_lds_floats(int NL, int NT1) {
    return (size_t)2 * NL * kSlabF + (size_t)(NL - 1) * kSlabF + (size_t)2 * kSR * ldx_of(NT1) + 2 * kSR + kHP + kAL + 256 * 4;
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
    int maxtiles;
    static constexpr int ntl(int l) { return l == 0 ? NT1 : 7; }
    constexpr TileMap() : tl{}, tn{}, tm{}, slot{}, cnt{}, lst{}, ntiles{}, maxtiles(0) {
        int id = 0;
        for (int l = 0; l < NL; ++l)
            for (int n = 0; n < ntl(l); ++n)
                for (int m = 0; m < 7; ++m) { tl[id] = l; tn[id] = n; tm[id] = m; ++id; }
        bool taken[NT] = {};
        // chain phase c: the waves without a chain tile (7 .. kBW-1) each multiply a 3 x 2 block of dW_{NL-1-c}, whose dZ is already
        // available: block e = wave - 7 -> in-tile pair {ntl-2-2(e/2), ntl-1-2(e/2)} x out-tile triple {3(e%2) .. 3(e%2)+2}
        for (int c = 0; c < CH; ++c) {
            const int l = NL - 1 - c;
            for (int v = 7; v < kBW; ++v) {
                const int e = v - 7, n0 = ntl(l) - 2 - 2 * (e / 2), m0 = 3 * (e % 2);
                for (int t = 0; t < NT; ++t)
                    if (tl[t] == l && tn[t] >= n0 && tn[t] < n0 + 2 && tm[t] >= m0 && tm[t] < m0 + 3 && n0 >= 0) { taken[t] = true; lst[v][c][cnt[v][c]++] = t; }
            }
        }
        // all-wave phase: waves 0..6 (which also hold the chain weights) get layer-pure runs of `take[l]` tiles in (nt, mt) order
        // (at most 7 dZ fragments + 3-4 A fragments per k-step); wave 7 collects every layer's leftover
        int T[NL] = {}, k[NL] = {}, take[NL] = {}, left[NL] = {};
        int rem = 0;
        for (int t = 0; t < NT; ++t)
            if (!taken[t]) { ++T[tl[t]]; ++rem; }
        const int base = rem / kBW;                  // wave 7 (no chain weights in registers) may end up with more accumulator tiles
        const int quota7 = base;
        // issue cost of a tile per k-step: a 16x16x4 MFMA 32 cycles, the 4x4x1 16-block form of the strip tiles (out-feature tile 6, and
        // in-feature tile 6 of the hidden layers) about 14 (scratch/mfma4) — the runs are cut by COST, not by tile count
        int C[NL] = {}, Ctot = 0;
        for (int t = 0; t < NT; ++t)
            if (!taken[t]) { const int c = (tm[t] == 6 || (tl[t] >= 1 && tn[t] == 6)) ? 14 : 32; C[tl[t]] += c; Ctot += c; }
        const int budget = Ctot / kBW;
        int ksum = 0;
        for (int l = 0; l < NL; ++l) { k[l] = base > 0 ? T[l] / base : 0; ksum += k[l]; }
        while (ksum > kBW - 1) {                       // too many runs: drop one from the layer with the smallest remainder
            int best = -1;
            for (int l = 0; l < NL; ++l)
                if (k[l] > 0 && (best < 0 || T[l] - k[l] * base < T[best] - k[best] * base)) best = l;
            --k[best]; --ksum;
        }
        while (ksum < kBW - 1) {                       // too few: add one to the layer with the largest leftover
            int best = 0;
            for (int l = 1; l < NL; ++l)
                if (T[l] - k[l] * base > T[best] - k[best] * base) best = l;
            ++k[best]; ++ksum;
        }
        int L = 0;
        (void)quota7;
        for (int l = 0; l < NL; ++l) {
            // tiles per run so that a run costs about one wave's share of the phase (average tile cost of the layer = C / T)
            int tk = k[l] > 0 && C[l] > 0 ? (budget * T[l]) / C[l] : 0;
            if (k[l] > 0 && tk > T[l] / k[l]) tk = T[l] / k[l];
            take[l] = tk; left[l] = T[l] - k[l] * take[l]; L += left[l];
        }
        {
            int w = 0;
            for (int l = 0; l < NL; ++l) {
                int given = 0, run = 0;
                for (int t = 0; t < NT; ++t) {
                    if (taken[t] || tl[t] != l) continue;
                    if (run < k[l]) {
                        lst[w][CH][cnt[w][CH]++] = t;
                        if (++given == take[l]) { given = 0; ++run; ++w; }
                    } else {
                        lst[kBW - 1][CH][cnt[kBW - 1][CH]++] = t;
                    }
                }
            }
        }
        for (int v = 0; v < kBW; ++v) {
            int s = 0;
            for (int ph = 0; ph < NPH; ++ph)
                for (int k = 0; k < cnt[v][ph]; ++k) slot[lst[v][ph][k]] = s++;
            ntiles[v] = s;
            if (s > maxtiles) maxtiles = s;
        }
    }
    // first use of the dZ fragment (l, mt) / the A fragment (l, nt) inside a wave's phase list => it has to be read from LDS
    constexpr bool first_a(int w, int ph, int k) const {
        const int t = lst[w][ph][k];
        for (int i = 0; i < k; ++i) { const int u = lst[w][ph][i]; if (tl[u] == tl[t] && tm[u] == tm[t]) return false; }
        return true;
    }
    // hidden-layer tiles (l >= 1, n == 6, m < 6): in-features 96..99 (+ the ones column) against 16 out-features — the "n strip"
    constexpr bool is_nstrip(int t) const { return tl[t] >= 1 && tn[t] == 6 && tm[t] < 6; }
    constexpr bool first_nstrip(int w, int ph, int k) const {
        const int t = lst[w][ph][k];
        if (!is_nstrip(t)) return false;
        for (int i = 0; i < k; ++i) { const int u = lst[w][ph][i]; if (tl[u] == tl[t] && is_nstrip(u)) return false; }
        return true;
    }
    constexpr bool first_b(int w, int ph, int k) const {
        const int t = lst[w][ph][k];
        if (is_nstrip(t)) return false;                       // the strip form reads its own 4-feature fragment
        for (int i = 0; i < k; ++i) { const int u = lst[w][ph][i]; if (tl[u] == tl[t] && tn[u] == tn[t] && !is_nstrip(u)) return false; }
        return true;
    }
};
template <int NL, int NT1> inline constexpr TileMap<NL, NT1> kTM{};

using lds_f = __attribute__((address_space(3))) float;
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
        {                                                                                                               \
            unsigned long long *tr_ = reinterpret_cast<unsigned long long *>(ws + (size_t)gridDim.x * np_stride) + (nslab_done * kBW + W) * 16; \
            tr_[(i)] = clock64();                                                                                       \
            if ((i) == 0) tr_[15] = wall_clock64();          /* 100 MHz real-time counter at slab start */              \
        }                                                                                                               \
    } while (0)
#else
#define BWD_STAMP(i) do { } while (0)
#endif

// Per slab s (parity p) a workgroup runs three barrier-separated phases; the LDS images of slab s+1 are prepared while slab s
// is multiplied, by the half of the waves that is NOT on the matrix pipe at that moment (waves w and w+4 share a SIMD):
//   A  chain 0 (waves 0-6; wave 7: its dW block) | waves 0-3 first issue the LDS-DMA + X loads of slab s+1, waves 4-7 issue
//      their DMA share after their MFMAs
//   B  chain 1 ... (NL-1 chain phases in general); then the DMA has landed (vmcnt(0)) and dLoss/dscore of slab s+1 is published
//   C  all remaining dW tiles of slab s | waves 0-3 first turn their X registers into the XS image of slab s+1 (input dropout
//      recomputed), waves 4-7 run the top-layer pass of slab s+1 (dZ in place, d w_out, d b_out, db_top) after their MFMAs
// Every wave runs its own specialisation of this body (tile lists, accumulator registers and roles are compile-time per wave).
template <int NL, int NT1, int W>
__device__ __forceinline__ void bwd_body(const float *__restrict__ X, const float *__restrict__ P, const float *__restrict__ acts,
                                         const float *__restrict__ dpreds, const MlpArgs a, float *__restrict__ ws, size_t np_stride,
                                         float *smem, float *__restrict__ dz0) {
    using TMt = TileMap<NL, NT1>;
    constexpr int CH = NL - 1, LDX = ldx_of(NT1), NTMAX = NT1 > 7 ? NT1 : 7;
    constexpr int NACC = kTM<NL, NT1>.ntiles[W] > 0 ? kTM<NL, NT1>.ntiles[W] : 1;
    // NT1 == 0 is the TAIL form for inputs too wide for the first layer's accumulators (r4): no X image, no dW_0 tiles; the chain's last
    // image dZ_0 goes to HBM (dz0) for the separate wide-input dW kernel (scorer_dw_x6.hip), which also takes db_0 off its column sums
    constexpr bool TAIL = NT1 == 0;
    constexpr bool loader = W < 4;                        // X staging + early DMA issue; the others: late DMA issue
    constexpr bool topper = W >= 4 && W < 8;              // top-layer pass (252 threads of waves 4..7)
    const int F = a.F, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    float *bufA = smem;                                   // [2][NL][kSlabF]  stored activations (top layer: dZ in place)
    float *Zb = bufA + 2 * NL * kSlabF;                   // [CH][kSlabF]     dZ_0 .. dZ_{NL-2}
    float *XSb = Zb + CH * kSlabF;                        // [2][kSR][LDX]    dropped input features
    float *dsb = XSb + 2 * kSR * LDX;                     // [2][kSR]         dLoss/dscore
    float *wo = dsb + 2 * kSR;                            // [kHP]            w_out
    float *junk_row = wo + kHP;                           // [kAL]            image row for the idle slots of the top-layer pass
    float *junk_x = junk_row + kAL;                       // [256][4]         store target of the idle X-staging slots
    const uint32_t bufA_addr = lds_byte_addr(bufA);
    const int nslabs = (R + kSR - 1) / kSR;
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    // ---- one-time: w_out to LDS, this wave's W^T fragments to registers
    for (int i = tid; i < kHP; i += kBT) { wo[i] = i < kH ? P[off_wout(NL, F) + i] : 0.0f; junk_row[i] = 0.0f; }
    float wf[CH > 0 ? CH : 1][25];
    static_for<CH>([&](auto c_) {
        constexpr int c = c_, l = NL - 1 - c;
        const float *Wl = P + off_W(l, F);
        const int k = 16 * W + j;
        const bool kok = W < 7 && k < kH;
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
    // consume the fragments HERE: hipcc otherwise sinks the waits for these loads to their first use inside the slab loop, as
    // s_waitcnt vmcnt(N) with N counted without the LDS-DMA in flight there — every slab would wait for its own prefetch
#pragma unroll
    for (int c = 0; c < (CH > 0 ? CH : 1); ++c)
#pragma unroll
        for (int i = 0; i < 25; ++i) asm volatile("" : "+v"(wf[c][i]));

    // ---- X staging (waves 0-3, 256 threads): slot u of thread t -> (slab row, float4 column), recomputed at use
    constexpr int XTOT = kSR * 4 * NT1, XSL = (XTOT + 255) / 256, XW4 = TAIL ? 1 : 4 * NT1;
    auto x_slot = [&](int t, int u, int &xr, int &xc) -> bool {
        const int idx = t + 256 * u;
        const bool in = idx < XTOT;
        xr = in ? idx / XW4 : 0;
        xc = in ? idx - xr * XW4 : 0;
        return in;
    };
    f32x4 xraw[(loader && XSL > 0) ? XSL : 1];
    float dsraw = 0.0f;

    // this wave's share of the activation DMA (chunks of 64 lanes x 16 B)
    auto issue_dma = [&](int slab, int buf, int lane_o) {
        // a slab image is CONTIGUOUS in global memory: two row tiles of 7 KB each (tile-major, ptr_mlp.h), chunk ch < 7 = feature tile ch of the
        // first, ch >= 7 of the second; a tail slab whose second row tile does not exist re-reads the last one (rows >= R only need finite
        // values: their dLoss/dscore is 0) — all scalar arithmetic
        const int nrt = act_row_tiles(R);
#pragma unroll
        for (int q = 0; q < (NL * 14 + kBW - 1) / kBW; ++q) {
            const int k = W + kBW * q;                     // scalar
            if (k < NL * 14) {
                const int layer = k / 14, ch = k - 14 * layer;
                const int rt = min(2 * slab + (ch >= 7 ? 1 : 0), nrt - 1);
                const float *src = acts + (size_t)layer * act_layer_floats(R) + (size_t)rt * kActTile + (ch >= 7 ? ch - 7 : ch) * 256 + 4 * lane_o;
                const uint32_t dst = bufA_addr + (uint32_t)(((buf * NL + layer) * kSlabF + ch * 256) * 4);
                glds16(src, __builtin_amdgcn_readfirstlane(dst));
            }
        }
    };
    auto load_x = [&](int slab, int tid_o) {              // raw loads, no consumer until finish_x (waves 0-3); dLoss/dscore: every wave
        const int row0 = slab * kSR;
        if constexpr (loader) {
#pragma unroll
            for (int u = 0; u < XSL; ++u) {
                int xr, xc;
                x_slot(tid_o, u, xr, xc);
                const int gr = min(row0 + xr, R - 1);
                const int cc = 4 * xc < F ? 4 * xc : 0;
                xraw[u] = *reinterpret_cast<const f32x4 *>(X + (size_t)gr * F + cc);
            }
        }
        dsraw = dpreds[min(row0 + (tid_o & (kSR - 1)), R - 1)];
    };
    auto finish_x = [&](int slab, int buf, int tid_o) {
      if constexpr (loader) {
        const int row0 = slab * kSR;
        float *XS = XSb + buf * kSR * LDX;
        const uint32_t thr_e = a.p_drop > 0.0f ? thr : 0u;          // threshold 0 keeps everything: no branch around the hash, so the
        // XSL slots form ONE basic block and their (serially dependent) hash chains interleave
#pragma unroll
        for (int u = 0; u < XSL; ++u) {
            int xr, xc;
            const bool in = x_slot(tid_o, u, xr, xc);
            uint32_t w0, w1;
#ifdef BWD_NO_HASH                                                       // timing experiment only (wrong masks)
            w0 = w1 = 0xFFFFFFFFu;
#else
            drop_bits(a.seed_lo, a.seed_hi, 0, min(row0 + xr, R - 1), xc, w0, w1);
#endif
            f32x4 v = drop4(xraw[u], w0, w1, thr_e, inv_keep);
            const float okf = 4 * xc < F ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= okf;
            v[0] = 4 * xc == F ? 1.0f : v[0];                        // the ones column: dW_0's column F is db_0
            float *dst = in ? XS + xr * LDX + 4 * xc : junk_x + 4 * tid_o;      // idle slots store to a pad instead of branching
            *reinterpret_cast<f32x4 *>(dst) = v;
        }
      }
    };
    // top layer (waves 4-7, 252 threads = 9 row slots x 28 feature groups): dZ_top = ds * w_out * [h > 0] in place,
    // d w_out += h * ds, d b_out += ds
    f32x4 dwo4 = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.0f;
    // Bias gradients of the hidden layers 1..NL-1 are column sums of their dZ images, taken where the images are PRODUCED (the chain
    // epilogue for dZ_1..dZ_{NL-2}, the top-layer pass for dZ_{NL-1}); only db_0 still comes off the ones column of the X image.
    f32x4 dbc[CH > 1 ? CH - 1 : 1];
#pragma unroll
    for (int i = 0; i < (CH > 1 ? CH - 1 : 1); ++i) dbc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dbt4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto top_pass = [&](int buf, int tid_o) {
      if constexpr (topper) {
        const int t = tid_o - 256;
        const int rs = t / 28, f4 = t - rs * 28;
        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wo + 4 * f4);
        float *img = bufA + (buf * NL + NL - 1) * kSlabF;
        const float first = f4 == 0 ? 1.0f : 0.0f;
        // branch-free: slots without a row (thread >= 252, or row >= 32 in the last pass) work on a zeroed pad row with ds = 0
        float *pa[4];
        f32x4 h4[4];
        float ds[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {                      // all reads first: the in-place update must not serialise the four rows
            const int r = rs + 9 * h;
            const bool ok = rs < 9 && r < kSR;
            pa[h] = ok ? img + (r >> 4) * kActTile + (f4 >> 2) * 256 + (r & 15) * 16 + (f4 & 3) * 4 : junk_row + 4 * f4;
            h4[h] = *reinterpret_cast<const f32x4 *>(pa[h]);
            const float dsr = dsb[buf * kSR + (ok ? r : 0)];
            ds[h] = ok ? dsr : 0.0f;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x4 d;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d[c] = (ds[h] * w4[c]) * (h4[h][c] > 0.0f ? 1.0f : 0.0f);
                dwo4[c] = fmaf(h4[h][c], ds[h], dwo4[c]);
                dbt4[c] += d[c];                                      // db_{NL-1}: column sum of dZ_top
            }
            *reinterpret_cast<f32x4 *>(pa[h]) = d;
            dbo = fmaf(first, ds[h], dbo);
        }
      }
    };

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dW tiles of wave WV in phase PH over the slab's 8 k-steps (4 rows each), software-pipelined by hand: two k-steps per loop
    // iteration, the operands of the next k-step are read into the other register set while the current one is multiplied
    auto dw_phase = [&](auto wv_, auto ph_, const float *cur, const float *XS) {
        constexpr int WV = wv_, PH = ph_;
        constexpr int CNT = kTM<NL, NT1>.cnt[WV][PH];
        if constexpr (CNT > 0) {
            float av[2][NL][kMT], bv[2][NL][NTMAX], bv6[2][NL];
            // One OPAQUE base address per LDS image (lane part included): every read below is base + a small constant that fits the
            // ds_read offset field.  Without it each read address (image offset > 64 KB + constant) is a loop-invariant VGPR of its
            // own, hoisted out of the slab loop by the hundred and spilled to scratch.
            uint32_t zb[NL], ab[NL], zb6[NL], ab6[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                zb[l] = lds_byte_addr(l == NL - 1 ? cur + (NL - 1) * kSlabF : Zb + l * kSlabF) + (uint32_t)((g * 16 + j) * 4);
                ab[l] = l == 0 ? lds_byte_addr(XS) + (uint32_t)((g * LDX + j) * 4)
                               : lds_byte_addr(cur + (l - 1) * kSlabF) + (uint32_t)((g * 16 + j) * 4);
                zb6[l] = zb[l] - (uint32_t)((j - (j & 3)) * 4);       // out-feature tile 6 as 4x4 blocks: lane j reads feature 96 + (j & 3)
                ab6[l] = ab[l] - (uint32_t)((j - (j & 3)) * 4);       // in-feature tile 6 of a hidden layer likewise
                asm volatile("" : "+v"(zb[l]), "+v"(ab[l]), "+v"(zb6[l]), "+v"(ab6[l]));
            }
            auto rd = [&](int ks, auto set_) {
                constexpr int set = set_;
                static_for<CNT>([&](auto k_) {
                    constexpr int t = kTM<NL, NT1>.lst[WV][PH][k_];
                    constexpr int l = kTM<NL, NT1>.tl[t], n = kTM<NL, NT1>.tn[t], m = kTM<NL, NT1>.tm[t];
                    // image rows 4 ks + g of a tile-major image: row tile ks >> 2, row (4 (ks & 3) + g) in it — the lane part g * 16 + j sits in the base
                    const int kso = (ks >> 2) * kActTile + (ks & 3) * 64;
                    if constexpr (kTM<NL, NT1>.first_a(WV, PH, k_))
                        av[set][l][m] = *reinterpret_cast<lds_f *>((uintptr_t)((m == kMT - 1 ? zb6[l] : zb[l]) + (uint32_t)((kso + 256 * m) * 4)));
                    if constexpr (kTM<NL, NT1>.first_b(WV, PH, k_))
                        bv[set][l][n] = *reinterpret_cast<lds_f *>((uintptr_t)(ab[l] + (uint32_t)((l == 0 ? 4 * ks * LDX + 16 * n : kso + 256 * n) * 4)));
                    if constexpr (kTM<NL, NT1>.first_nstrip(WV, PH, k_))
                        bv6[set][l] = *reinterpret_cast<lds_f *>((uintptr_t)(ab6[l] + (uint32_t)((kso + 256 * 6) * 4)));
                });
            };
            auto mma = [&](auto set_) {
                constexpr int set = set_;
                static_for<CNT>([&](auto k_) {
                    constexpr int t = kTM<NL, NT1>.lst[WV][PH][k_];
                    constexpr int l = kTM<NL, NT1>.tl[t], n = kTM<NL, NT1>.tn[t], m = kTM<NL, NT1>.tm[t], s = kTM<NL, NT1>.slot[t];
                    // Out-feature tile 6 holds features 96..99 only: v_mfma_f32_4x4x1_16b_f32 — 16 independent 4 x 4 x 1 blocks, block
                    // b = 4 g + j / 4 = (document row 4 ks + g) x (in-features 16 n + 4 (j / 4) .. + 3), A = dZ[row][96 + i], B = the SAME
                    // fragment the 16x16 tiles of column n use — multiplies exactly the 4 x 16 x 4 useful products of the tile in 1/2..1/3 of
                    // a 16x16x4's issue time (scratch/mfma4: 12-16 cycles vs 32).  Lane (j, g) accumulates dW[96 + c][16 n + j] over the rows
                    // = g (mod 4); the four lane groups are added in the epilogue.
                    if constexpr (m == kMT - 1) {
                        acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][l][m], bv[set][l][n], acc[s], 0, 0, 0);
                    } else if constexpr (kTM<NL, NT1>.is_nstrip(t)) {
                        // In-feature tile 6 of a hidden layer holds features 96..99 and the ones column: the same 4x4 form with the roles
                        // swapped — A = the 4 in-features, B = the dZ fragment of the tile — lane (j, g) accumulates dW[16 m + j][96 + c] (the
                        // bias gradient does not need the ones column: see dbc / dbt4)
                        acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(bv6[set][l], av[set][l][m], acc[s], 0, 0, 0);
                    } else {
                        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][l][m], bv[set][l][n], acc[s], 0, 0, 0);
                    }
                });
            };
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            rd(0, S0{});
#pragma unroll
            for (int ks = 0; ks < kSR / 4; ks += 2) {
                rd(ks + 1, S1{});
                mma(S0{});
                rd(ks + 2 < kSR / 4 ? ks + 2 : ks + 1, S0{});      // last iteration: a harmless re-read instead of a branch
                mma(S1{});
            }
        }
    };
    int slab = blockIdx.x, p = 0;
    int nslab_done = 0;
    (void)nslab_done;
    // ---- prologue: slab 0's images
    issue_dma(slab, 0, lane);
    load_x(slab, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const float dsv = slab * kSR + (tid & (kSR - 1)) < R ? dsraw : 0.0f;
        if (tid < kSR) dsb[tid] = dsv;
    }
    if (loader) finish_x(slab, 0, tid);
    wg_barrier();
    if (topper) top_pass(0, tid);
    wg_barrier();

    for (; slab < nslabs; slab += gridDim.x, p ^= 1) {
        // The preparation of the next slab is UNCONDITIONAL (the last iteration redundantly re-prepares its own slab into the other
        // buffers): loads issued and consumed under a run-time `has_next` look pending to hipcc's waitcnt pass on the impossible
        // issue-but-skip-use path, and the s_waitcnt vmcnt(N) it then places at the next reuse of those registers lands right
        // behind freshly issued LDS-DMA (which it does not count) — a full HBM round trip per slab in the chain phase.
        const bool real_next = slab + gridDim.x < nslabs;
        const int next = real_next ? slab + gridDim.x : slab;        // redundant pass: its dLoss/dscore is published as 0 (no double count)
        float *cur = bufA + p * NL * kSlabF;
        // opaque copies: address arithmetic derived from them is recomputed per slab instead of being hoisted out of the loop
        // and kept in registers across the MFMA phases
        int tid_o = tid, lane_o = lane;
        asm volatile("" : "+v"(tid_o), "+v"(lane_o));
        BWD_STAMP(0);

        // ---- phases A, B, ...: chain (waves 0-6) / wave 7's dW blocks
        static_for<CH>([&](auto c_) {
            constexpr int c = c_, l = NL - 1 - c;
            if constexpr (c == 0) {
                if constexpr (loader) { PREP_BEGIN(); issue_dma(next, p ^ 1, lane_o); load_x(next, tid_o); PREP_END(); }
            }
            if constexpr (W < 7) {
                const float *src = c == 0 ? cur + (NL - 1) * kSlabF : Zb + l * kSlabF;
                float *dst = Zb + (l - 1) * kSlabF;
                const float *gate = cur + (l - 1) * kSlabF;
                // one row tile at a time (24 + 1 operand registers); two accumulators over alternating k-steps keep two
                // independent MFMA chains in flight (dependent-accumulator latency 40 cycles vs 32 issue)
#pragma unroll
                for (int rt = 0; rt < kSR / 16; ++rt) {
                    const int rbase = rt * kActTile + j * 16;         // document j of row tile rt: feature tile S at + 256 S (tile-major images)
                    f32x4 b[6];
#pragma unroll
                    for (int S = 0; S < 6; ++S) b[S] = *reinterpret_cast<const f32x4 *>(src + rbase + 256 * S + 4 * g);
                    const float bt = src[rbase + 256 * 6 + g];
                    const f32x4 gt = *reinterpret_cast<const f32x4 *>(gate + rbase + 256 * W + 4 * g);
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
                    *reinterpret_cast<f32x4 *>(dst + rbase + 256 * W + 4 * g) = d;
                    if constexpr (c < CH - 1) dbc[c] += d;            // db_{l-1}: column sum of the dZ image just produced (rows >= R are 0)
                }
            } else {
                dw_phase(std::integral_constant<int, W>{}, c_, cur, XSb + p * kSR * LDX);
            }
            if constexpr (c == 0) {
                if constexpr (!loader) { PREP_BEGIN(); issue_dma(next, p ^ 1, lane_o); load_x(next, tid_o); PREP_END(); }
            }
            if constexpr (c == CH - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // slab s+1's images have landed
                const float dsv = (real_next && next * kSR + (tid_o & (kSR - 1)) < R) ? dsraw : 0.0f;
                if (tid_o < kSR) dsb[(p ^ 1) * kSR + tid_o] = dsv;
            }
            BWD_STAMP(1 + 2 * c);
            wg_barrier();
            BWD_STAMP(2 + 2 * c);
        });

        // ---- phase C: all-wave dW of slab s; preparation of slab s+1 by the wave half that is off the matrix pipe
        if constexpr (loader && !TAIL) { PREP_BEGIN(); finish_x(next, p ^ 1, tid_o); PREP_END(); }
        if constexpr (loader && TAIL) {                  // the dZ_0 image of this slab (complete since the last chain barrier) -> dz0, rows < R
            const int row0 = slab * kSR;
#pragma unroll
            for (int u = 0; u < (kSR * (kAL / 4) + 255) / 256; ++u) {
                const int idx = tid_o + 256 * u;
                const int r = idx / (kAL / 4), c4 = idx - r * (kAL / 4);
                if (idx < kSR * (kAL / 4) && row0 + r < R)
                    *reinterpret_cast<f32x4 *>(dz0 + (size_t)(row0 + r) * kAL + 4 * c4) = *reinterpret_cast<const f32x4 *>(Zb + act_off(r, 4 * c4));   // dz0 is row-major
            }
        }
        BWD_STAMP(1 + 2 * CH);
        dw_phase(std::integral_constant<int, W>{}, std::integral_constant<int, CH>{}, cur, XSb + p * kSR * LDX);
        BWD_STAMP(2 + 2 * CH);
        if constexpr (topper) { PREP_BEGIN(); top_pass(p ^ 1, tid_o); PREP_END(); }
        BWD_STAMP(3 + 2 * CH);
        wg_barrier();
        BWD_STAMP(4 + 2 * CH);
        ++nslab_done;
    }

    // ---- epilogue: this workgroup's partial gradient in the flat parameter layout
    float *out = ws + (size_t)blockIdx.x * np_stride;
    auto store_tiles = [&](auto wv_) {
        constexpr int WV = wv_;
        static_for<TMt::NPH>([&](auto ph_) {
            constexpr int PH = ph_;
            static_for<kTM<NL, NT1>.cnt[WV][PH]>([&](auto k_) {
                constexpr int t = kTM<NL, NT1>.lst[WV][PH][k_];
                constexpr int l = kTM<NL, NT1>.tl[t], n = kTM<NL, NT1>.tn[t], m = kTM<NL, NT1>.tm[t], s = kTM<NL, NT1>.slot[t];
                const int K = l == 0 ? F : kH;
                const int k = 16 * n + j;
                float *o = out + off_W(l, F);
                if constexpr (m == kMT - 1) {
                    // 4x4 form: element c = out-feature 96 + c, partial over the document rows = g (mod 4): fixed-order sum over the lane groups
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = acc[s][c];
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        if (g == 0 && k < K) o[(size_t)(96 + c) * K + k] = v;
                        if (l == 0 && g == 0 && k == K) out[off_b(l, F) + 96 + c] = v;      // db_0: the ones column of the X image
                    }
                } else if constexpr (kTM<NL, NT1>.is_nstrip(t)) {
                    // element c = in-feature 96 + c of out-feature 16 m + j, partial over the document rows = g (mod 4)
                    const int of = 16 * m + j;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = acc[s][c];
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        if (g == 0) o[(size_t)of * K + 96 + c] = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int of = 16 * m + 4 * g + c;
                        if (of < kH && k < K) o[(size_t)of * K + k] = acc[s][c];
                        if (l == 0 && of < kH && k == K) out[off_b(l, F) + of] = acc[s][c];        // db_0: the ones column of the X image
                    }
                }
            });
        });
    };
    store_tiles(std::integral_constant<int, W>{});
    // d w_out / d b_out: fixed-order sums over the 9 row slots of the top-layer pass (Zb is free after the last barrier)
    if (topper) {
        const int t = tid - 256, rs = t / 28, f4 = t - rs * 28;
        if (rs < 9) {
            *reinterpret_cast<f32x4 *>(Zb + rs * kAL + 4 * f4) = dwo4;
            *reinterpret_cast<f32x4 *>(Zb + (9 + rs) * kAL + 4 * f4) = dbt4;
            if (f4 == 0) Zb[18 * kAL + rs] = dbo;
        }
    }
    // chain waves: db of the layers whose dZ they produced — fixed-order sum over the 16 document lanes of a lane group
    if constexpr (W < 7) {
        static_for<(CH > 1 ? CH - 1 : 0)>([&](auto c_) {
            constexpr int c = c_, lyr = NL - 2 - c;              // chain phase c produced dZ_{NL-2-c}
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = dbc[c][e];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                const int of = 16 * W + 4 * g + e;
                if (j == 0 && of < kH) out[off_b(lyr, F) + of] = v;
            }
        });
    }
    wg_barrier();
    if (tid < kH) {
        float s = 0.0f, sb = 0.0f;
#pragma unroll
        for (int r = 0; r < 9; ++r) { s += Zb[r * kAL + tid]; sb += Zb[(9 + r) * kAL + tid]; }
        out[off_wout(NL, F) + tid] = s;
        out[off_b(NL - 1, F) + tid] = sb;
    } else if (tid == kH) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 9; ++r) s += Zb[18 * kAL + r];
        out[off_wout(NL, F) + kH] = s;
    }
}

template <int NL, int NT1>
__global__ void __launch_bounds__(kBT)
mlp_bwd_fused_kernel(const float *__restrict__ X, const float *__restrict__ P, const float *__restrict__ acts,
                     const float *__restrict__ dpreds, MlpArgs a, float *__restrict__ ws, size_t np_stride, float *__restrict__ dz0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: bwd_body<NL, NT1, 0>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 1: bwd_body<NL, NT1, 1>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 2: bwd_body<NL, NT1, 2>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 3: bwd_body<NL, NT1, 3>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 4: bwd_body<NL, NT1, 4>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 5: bwd_body<NL, NT1, 5>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 6: bwd_body<NL, NT1, 6>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
#if BWD_WAVES == 12
        case 7: bwd_body<NL, NT1, 7>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 8: bwd_body<NL, NT1, 8>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 9: bwd_body<NL, NT1, 9>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        case 10: bwd_body<NL, NT1, 10>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
        default: bwd_body<NL, NT1, 11>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
#else
        default: bwd_body<NL, NT1, 7>(X, P, acts, dpreds, a, ws, np_stride, smem, dz0); break;
#endif
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
    // F < 16 * NT1: the X image needs a free column F for the ones column that makes column F of dW_0 the bias gradient
    // (F = 144 fills all nine tiles: it takes the layer-wise kernels)
    return NL == 3 && NT1 == 9 && F % 4 == 0 && F < 16 * NT1 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(acts) & 15) == 0;
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
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBT), lds, st, X, params, acts, dpreds, a, ws, NP, (float *)nullptr);
        return check_hip(hipGetLastError(), who);
    };
    return go(mlp_bwd_fused_kernel<3, 9>, 9);
}

// The TAIL form (NT1 = 0): chain + every hidden-layer gradient + d w_out / d b_out in one pass over the stored activations, dZ of the first
// layer written to dz0 [R][112] for the wide-input dW kernel.  PTR_BWD_TAIL=0 keeps the layer-wise dZ / dW kernels (A/B measurements, tests).
bool bwd_tail_supported(int NL, const void *acts) {
    const char *e = getenv("PTR_BWD_TAIL");
    if (e && atoi(e) == 0) return false;
    return (NL == 2 || NL == 3) && (reinterpret_cast<uintptr_t>(acts) & 15) == 0;
}

int launch_bwd_tail(const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, float *dz0, hipStream_t st,
                    const char *who) {
    const int grid = bwd_fused_grid(a.R);
    const size_t NP = n_params(a.NL, a.F);
    auto go = [&](auto kern) -> int {
        const size_t lds = bwd_fused_lds_floats(a.NL, 0) * sizeof(float);
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBT), lds, st, (const float *)nullptr, params, acts, dpreds, a, ws, NP, dz0);
        return check_hip(hipGetLastError(), who);
    };
    return a.NL == 2 ? go(mlp_bwd_fused_kernel<2, 0>) : go(mlp_bwd_fused_kernel<3, 0>);
}

}  // namespace ptr
