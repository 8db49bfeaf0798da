// Fused kernels of the listwise (permutation-equivariant) scorer `listsf`: the multi-head self-attention core on fp32 MFMA
// and the reference's own LayerNorm.
//
// Reference: ptranking/base/list_ranker.py:176-254 (MultiheadAttention.forward: Q K^T / sqrt(d_h) -> softmax -> Dropout -> . V,
//            heads are column blocks of the [B, L, F] projections), :152-174 (LayerNorm: a_2*(x-mean)/(std+eps)+b_2 with the
//            UNBIASED std and eps added to std, not to the variance).  Defaults parameter.py:152-166 (2 heads, 6 layers,
//            dropout 0.1).  The Linear projections around the core stay library GEMMs.
//
// The reference materialises scores, softmax, dropout mask and dropped softmax as four [B, H, L, L] fp32 tensors per layer
// (0.54 GB each at B = 1024, L = 256) and keeps them for backward.  Here nothing of size L^2 touches HBM:
//   forward   one workgroup (4 or 8 waves) per (query, head, block of 16*RT rows per wave): K/V stream through LDS in chunks of
//             8 keys per wave, software-pipelined (the next chunk's loads fly during the MFMAs); online softmax, dropout bits
//             from a counter hash; O and the per-row log-sum-exp are the only outputs;
//   backward  recomputes P = exp(S - lse) tile by tile: one kernel per row block for dQ, one per key block for dK / dV
//             (two kernels instead of atomics on dQ: every output element has one owner, so results are bit-stable).
// MFMA formulation (v_mfma_f32_16x16x4_f32, exact fp32), "transposed world" as in scorer.hip:
//   S^T[key][row] = K[key][:] . Q[row][:]      -> lane (j = l&15, g = l>>4) holds row j, keys 16*kt + 4*g + {0..3}
//   which IS the B-operand layout of   O^T[d][row] += V^T[d][key] * P^T[key][row]   (the k index may be visited in any
//   order as long as A and B agree), so probabilities never leave registers.  Contractions over d keep the wave's own operand
//   (its Q rows; its K / V rows in the dK/dV kernel) in registers for the whole kernel and read the streamed operand from
//   LDS as one ds_read_b128 per 4 k-steps (k-slot (c, g) <-> d = 16*blk + 4*g + c), shared by all row tiles.
#include <stdlib.h>

#include "ptr_device.h"
#include "ptr_dropout.h"

namespace ptr {

constexpr int kRC = 32;                // rows per LDS chunk (dK / dV)
constexpr int kDsPadLd = 20;           // row stride (floats) of the dS transpose pad: 16-byte aligned rows, 80 B = 20 banks apart
// workgroups per CU the attention kernels are compiled for (register budget): the backward kernels gain 4 % from a third wave per
// SIMD (<= 168 VGPRs), the forward loses 10 % (measured at 1024 x 256 x 136, 2 heads: scratch/exp_attn.py)
#ifndef PTR_ATTN_MINBLK_FWD
#define PTR_ATTN_MINBLK_FWD 2
#endif
#ifndef PTR_ATTN_MINBLK_BWD
#define PTR_ATTN_MINBLK_BWD (DT <= 5 ? 3 : 2)      // head dimensions above 80 would spill 80-170 registers at 168
#endif

struct AttnArgs {
    int B, L, H, dh, F;
    int ld;                            // row stride (floats) of Q / K / V and dQ / dK / dV: F, or 3F for a packed [B][L][3F] projection
    float inv_scale;                   // 1 / sqrt(dh)
    float p_drop;
    uint32_t seed_lo, seed_hi;
    int site;
};

// LDS leading dimension for a [rows][dh] tile: covers the 16*DT columns the d-tiles touch, ld/4 odd (conflict-free b128)
__host__ __device__ constexpr int attn_ld(int DT) { return ((16 * DT / 4) & 1) ? 16 * DT : 16 * DT + 4; }

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The row (key) blocks of one (query, head) re-read the
// same K / V (Q / dO) rows, so consecutive LOGICAL block ids are placed on the same XCD: logical = xcd * (n / 8) + slot.
__device__ __forceinline__ int xcd_major_block_id() {
    const int n = gridDim.x, b = blockIdx.x;
    return (n & 7) == 0 ? (b & 7) * (n >> 3) + (b >> 3) : b;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Software-pipelined staging of rows [row0, row0 + NROWS) of a head's column block into an LDS tile dst[NROWS][LD] (rows >=
// row_lim and columns >= dh zero): load() issues the global loads of the NEXT chunk into registers before the
// current chunk is consumed, store() commits them to LDS afterwards, so the HBM/L2 latency hides behind the MFMA work.
template <int NROWS, int LD, int NT>
struct RowStage {
    static constexpr int LD4 = LD / 4, TOT = NROWS * LD4, NIT = (TOT + NT - 1) / NT;
    f32x4 v[NIT];
    // Raw loads only (clamped, always-valid addresses): nothing here may consume the loaded values, or the s_waitcnt lands
    // in front of the compute phase the loads are supposed to hide behind.  store() applies the zero padding.
    __device__ __forceinline__ void load(const float *src, int F, int dh, int row0, int row_lim, int tid) {
        const bool vec = ((dh & 3) == 0) && ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
        const int last = row_lim > 0 ? row_lim - 1 : 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT;
            const int r = idx / LD4, c = (idx - r * LD4) << 2;
            const int row = row0 + r < row_lim ? row0 + r : last;
            if (vec) {
                v[it] = *reinterpret_cast<const f32x4 *>(src + (size_t)row * F + (c < dh ? c : 0));
            } else {
                const float *p = src + (size_t)row * F;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[it][e] = p[c + e < dh ? c + e : 0];
            }
        }
    }
    __device__ __forceinline__ void store(float *dst, int dh, int row0, int row_lim, int tid) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT;
            if (idx < TOT) {
                const int r = idx / LD4, c = (idx - r * LD4) << 2;
                const bool rok = row0 + r < row_lim;
                f32x4 t = v[it];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] *= (rok && c + e < dh) ? 1.0f : 0.0f;
                *reinterpret_cast<f32x4 *>(dst + (size_t)r * LD + c) = t;
            }
        }
    }
};

// One MFMA operand (16 rows x head dimension) held in registers in k-slot order: v[blk][c] <-> d = 16*blk + 4*g + c,
// t[i] <-> d = 16*nb + 4*i + g (the dh % 16 tail).  Loaded once from an LDS tile row (row = tile row of lane j).
template <int DT>
struct OperandRegs {
    f32x4 v[DT];
    float t[3];
    __device__ __forceinline__ void load(const float *row, int nb, int rem, int g) {
#pragma unroll
        for (int blk = 0; blk < DT; ++blk) v[blk] = blk < nb ? *reinterpret_cast<const f32x4 *>(row + 16 * blk + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = i < rem ? row[16 * nb + 4 * i + g] : 0.0f;
    }
    // Same operand straight from global memory (row = this lane's row of the head's column block; rows that do not exist
    // are passed as a valid clamped pointer with ok = false).  Columns >= dh read as zero.
    __device__ __forceinline__ void load_global(const float *row, bool ok, int dh, int nb, int rem, int g, bool vec) {
        const float m = ok ? 1.0f : 0.0f;
#pragma unroll
        for (int blk = 0; blk < DT; ++blk) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (blk < nb) {
                const int c = 16 * blk + 4 * g;
                if (vec) x = *reinterpret_cast<const f32x4 *>(row + c);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = row[c + e < dh ? c + e : 0] * (c + e < dh ? 1.0f : 0.0f);
                }
            }
            v[blk] = x * m;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int d = 16 * nb + 4 * i + g;
            t[i] = i < rem ? row[d < dh ? d : 0] * ((d < dh) ? m : 0.0f) : 0.0f;
        }
    }
};

// acc[ib][ia][r] = sum_d A[a_row0 + 16*ia + 4*g + r][d] * B_ib[j][d]:  NA A-tiles read from LDS (one ds_read_b128 per tile and
// 4 k-steps), NB B-operands from registers.  NA*NB independent accumulators keep the MFMA pipe free of dependent chains.
template <int DT, int NA, int NB>
__device__ __forceinline__ void multi_dot(const float *As, int a_row0, int ld, int nb, int rem, int j, int g,
                                          const OperandRegs<DT> (&b)[NB], f32x4 (&acc)[NB][NA]) {
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) acc[ib][ia] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap = As + (size_t)(a_row0 + j) * ld;
#pragma unroll
    for (int blk = 0; blk < DT; ++blk) {
        if (blk < nb) {
            f32x4 a[NA];
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) a[ia] = *reinterpret_cast<const f32x4 *>(ap + (size_t)16 * ia * ld + 16 * blk + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ia = 0; ia < NA; ++ia)
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) acc[ib][ia] = mfma4(a[ia][c], b[ib].v[blk][c], acc[ib][ia]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < rem) {
            const int d = 16 * nb + 4 * i + g;
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) {
                const float av = ap[(size_t)16 * ia * ld + d];
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) acc[ib][ia] = mfma4(av, b[ib].t[i], acc[ib][ia]);
            }
        }
    }
}

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// ============================================================================================ forward
template <int DT, int RT, int NW>
__global__ void __launch_bounds__(NW * 64, PTR_ATTN_MINBLK_FWD)
mhsa_fwd_kernel(const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V,
                const int32_t *__restrict__ lens, AttnArgs a, float *__restrict__ O, float *__restrict__ LSE) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ld = attn_ld(DT), RPB = 16 * RT * NW, NT = NW * 64, KC = 8 * NW, NKT = KC / 16;
    float *Ks = smem, *Vs = Ks + (size_t)KC * ld;
    const int L = a.L, F = a.F, dh = a.dh, ldi = a.ld;
    const int nrb = (L + RPB - 1) / RPB;
    const int lid = xcd_major_block_id();
    const int rb = lid % nrb, bh = lid / nrb, b = bh / a.H, h = bh - b * a.H;
    int n = lens ? lens[b] : L;
    n = n < 0 ? 0 : (n > L ? L : n);
    const size_t base = (size_t)b * L * ldi + (size_t)h * dh, obase = (size_t)b * L * F + (size_t)h * dh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int row0 = rb * RPB;
    const int dh4 = (dh + 3) & ~3, nb = dh4 >> 4, rem = (dh4 & 15) >> 2;
    const bool vecq = ((dh & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(Q + base) & 15) == 0);

    const uint32_t thr = drop_thr(a.p_drop);
    float m[RT], l[RT];
    f32x4 acc[RT][DT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m[rt] = -INFINITY; l[rt] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wrow = wave * 16 * RT;                       // first row of this wave inside the block
    OperandRegs<DT> qr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = row0 + wrow + 16 * rt + j;
        qr[rt].load_global(Q + base + (size_t)(row < L ? row : L - 1) * ldi, row < L, dh, nb, rem, g, vecq);
    }

    RowStage<KC, ld, NT> kst, vst;
    kst.load(K + base, ldi, dh, 0, n, tid);
    vst.load(V + base, ldi, dh, 0, n, tid);
    for (int kc = 0; kc < n; kc += KC) {
        __syncthreads();                                   // every wave is done with the previous chunk
        kst.store(Ks, dh, kc, n, tid);
        vst.store(Vs, dh, kc, n, tid);
        __syncthreads();
        if (kc + KC < n) {                                // prefetch the next chunk while this one is consumed
            kst.load(K + base, ldi, dh, kc + KC, n, tid);
            vst.load(V + base, ldi, dh, kc + KC, n, tid);
        }
        f32x4 p[RT][NKT];
        multi_dot<DT, NKT, RT>(Ks, 0, ld, nb, rem, j, g, qr, p);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kc + 16 * kt + 4 * g + r;
                    const float v = key < n ? p[rt][kt][r] * a.inv_scale : -INFINITY;       // list_ranker.py:223
                    p[rt][kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = xor_max(mx);
            const float m_new = fmaxf(m[rt], mx);
            const float corr = __expf(m[rt] - m_new);
            float rs = 0.0f;
            const int grow = bh * L + row0 + wrow + 16 * rt + j;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __expf(p[rt][kt][r] - m_new); rs += e[r]; }
                if (thr != 0) {                                                             // list_ranker.py:229
                    uint32_t w0, w1;
                    drop_bits(a.seed_lo, a.seed_hi, a.site, grow, (kc + 16 * kt + 4 * g) >> 2, w0, w1);
                    e = drop4(e, w0, w1, thr, 1.0f);
                }
                p[rt][kt] = e;
            }
            rs = xor_sum(rs);
            l[rt] = l[rt] * corr + rs;
            m[rt] = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[rt][dt] *= corr;
        }
        // O^T[d][row] += V^T[d][key] * P^T[key][row]                                          (list_ranker.py:236)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *vrow = Vs + (size_t)(16 * kt + 4 * g + r) * ld + j;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float va = vrow[16 * dt];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][dt] = mfma4(va, p[rt][kt][r], acc[rt][dt]);
                }
            }
        }
    }
    const float keep_inv = thr != 0 ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const bool vec = ((dh & 3) == 0) && ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(O + obase) & 15) == 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = row0 + wrow + 16 * rt + j;
        if (row >= L) continue;
        const float inv = l[rt] > 0.0f ? keep_inv / l[rt] : 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * g;
            const f32x4 o = acc[rt][dt] * inv;
            float *dst = O + obase + (size_t)row * F + d;
            if (vec) { if (d < dh) *reinterpret_cast<f32x4 *>(dst) = o; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (d + e < dh) dst[e] = o[e];
            }
        }
        if (g == 0) LSE[(size_t)bh * L + row] = l[rt] > 0.0f ? m[rt] + __logf(l[rt]) : 0.0f;
    }
}

// D[bh][row] = sum_d dO[row][d] * O[row][d]  (= sum_key P[row][key] * dP[row][key], also under dropout)
__global__ void __launch_bounds__(256)
attn_rowdot_kernel(const float *__restrict__ O, const float *__restrict__ dO, AttnArgs a, float *__restrict__ Dv) {
    const int lane = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const size_t nrows = (size_t)a.B * a.H * a.L;
    for (size_t r = (size_t)blockIdx.x * 16 + slot; r < nrows; r += (size_t)gridDim.x * 16) {
        const int bh = (int)(r / a.L), row = (int)(r - (size_t)bh * a.L), b = bh / a.H, h = bh - b * a.H;
        const size_t off = ((size_t)b * a.L + row) * a.F + (size_t)h * a.dh;
        float s = 0.0f;
        for (int d = lane; d < a.dh; d += 16) s += O[off + d] * dO[off + d];
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        if (lane == 0) Dv[r] = s;
    }
}

// ============================================================================================ backward: dQ
template <int DT, int NW>
__global__ void __launch_bounds__(NW * 64, PTR_ATTN_MINBLK_BWD)
mhsa_bwd_dq_kernel(const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V,
                   const float *__restrict__ dO, const float *__restrict__ LSE, const float *__restrict__ Dv,
                   const int32_t *__restrict__ lens, AttnArgs a, float *__restrict__ dQ) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ld = attn_ld(DT), RPB = 16 * NW, NT = NW * 64, KC = 8 * NW, NKT = KC / 16;
    float *Ks = smem, *Vs = Ks + (size_t)KC * ld;
    const int L = a.L, F = a.F, dh = a.dh, ldi = a.ld;
    const int nrb = (L + RPB - 1) / RPB;
    const int lid = xcd_major_block_id();
    const int rb = lid % nrb, bh = lid / nrb, b = bh / a.H, h = bh - b * a.H;
    int n = lens ? lens[b] : L;
    n = n < 0 ? 0 : (n > L ? L : n);
    const size_t base = (size_t)b * L * ldi + (size_t)h * dh, obase = (size_t)b * L * F + (size_t)h * dh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int row0 = rb * RPB, wrow = wave * 16;
    const int dh4 = (dh + 3) & ~3, nb = dh4 >> 4, rem = (dh4 & 15) >> 2;
    const int row = row0 + wrow + j;
    const bool rok = row < L;
    const float lse = rok ? LSE[(size_t)bh * L + row] : 0.0f;
    const float Dr = rok ? Dv[(size_t)bh * L + row] : 0.0f;
    const uint32_t thr = drop_thr(a.p_drop);
    const float keep_inv = thr != 0 ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    OperandRegs<DT> qr[1], gr[1];
    {
        const bool vq = ((dh & 3) == 0) && ((F & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(Q + base) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(dO + obase) & 15) == 0);
        const size_t rr = (size_t)(rok ? row : L - 1);
        qr[0].load_global(Q + base + rr * ldi, rok, dh, nb, rem, g, vq);
        gr[0].load_global(dO + obase + rr * F, rok, dh, nb, rem, g, vq);
    }

    RowStage<KC, ld, NT> kst, vst;
    kst.load(K + base, ldi, dh, 0, n, tid);
    vst.load(V + base, ldi, dh, 0, n, tid);
    for (int kc = 0; kc < n; kc += KC) {
        __syncthreads();
        kst.store(Ks, dh, kc, n, tid);
        vst.store(Vs, dh, kc, n, tid);
        __syncthreads();
        if (kc + KC < n) {
            kst.load(K + base, ldi, dh, kc + KC, n, tid);
            vst.load(V + base, ldi, dh, kc + KC, n, tid);
        }
        f32x4 ds[NKT], s4[1][NKT], dp4[1][NKT];
        multi_dot<DT, NKT, 1>(Ks, 0, ld, nb, rem, j, g, qr, s4);
        multi_dot<DT, NKT, 1>(Vs, 0, ld, nb, rem, j, g, gr, dp4);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const f32x4 s = s4[0][kt], dp = dp4[0][kt];
            f32x4 keep = {keep_inv, keep_inv, keep_inv, keep_inv};
            if (thr != 0) {
                uint32_t w0, w1;
                drop_bits(a.seed_lo, a.seed_hi, a.site, bh * L + row, (kc + 16 * kt + 4 * g) >> 2, w0, w1);
                keep = drop4(keep, w0, w1, thr, 1.0f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kc + 16 * kt + 4 * g + r;
                const float pr = (key < n && rok) ? __expf(s[r] * a.inv_scale - lse) : 0.0f;
                ds[kt][r] = pr * (dp[r] * keep[r] - Dr) * a.inv_scale;
            }
        }
        // dQ^T[d][row] += K^T[d][key] * dS^T[key][row]
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *krow = Ks + (size_t)(16 * kt + 4 * g + r) * ld + j;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = mfma4(krow[16 * dt], ds[kt][r], dq[dt]);
            }
        }
    }
    if (!rok) return;
    const bool vec = ((dh & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(dQ + base) & 15) == 0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        float *dst = dQ + base + (size_t)row * ldi + d;
        if (vec) { if (d < dh) *reinterpret_cast<f32x4 *>(dst) = dq[dt]; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (d + e < dh) dst[e] = dq[dt][e];
        }
    }
}

// ============================================================================================ backward: dQ from stored dS
// dQ^T[d][row] = sum_key K^T[d][key] * dS^T[key][row] with dS written by mhsa_bwd_dkv_kernel: one GEMM unit instead of three (the
// recomputing kernel above also forms S = Q K^T and dP = dO V^T).  Lane (j, g) = row j; its dS fragment for key tile kt is the float4
// dS[row][kc + 16 kt + 4 g ..] — the B operand layout directly.  Keys >= n (padding; columns a non-live key block never wrote) are
// SELECTED to zero, not multiplied.
template <int DT, int NW>
__global__ void __launch_bounds__(NW * 64, PTR_ATTN_MINBLK_BWD)
mhsa_bwd_dq_ds_kernel(const float *__restrict__ K, const float *__restrict__ dS_ws, const int32_t *__restrict__ lens, AttnArgs a,
                      float *__restrict__ dQ) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ld = attn_ld(DT), RPB = 16 * NW, NT = NW * 64, KC = 8 * NW, NKT = KC / 16;
    float *Ks = smem;
    const int L = a.L, dh = a.dh, ldi = a.ld;
    const int nrb = (L + RPB - 1) / RPB;
    const int lid = xcd_major_block_id();
    const int rb = lid % nrb, bh = lid / nrb, b = bh / a.H, h = bh - b * a.H;
    int n = lens ? lens[b] : L;
    n = n < 0 ? 0 : (n > L ? L : n);
    const size_t base = (size_t)b * L * ldi + (size_t)h * dh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int row = rb * RPB + wave * 16 + j;
    const bool rok = row < L;
    const float *dsrow = dS_ws + ((size_t)bh * L + (rok ? row : L - 1)) * L;
    const bool vds = (L & 3) == 0 && (reinterpret_cast<uintptr_t>(dS_ws) & 15) == 0;
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    RowStage<KC, ld, NT> kst;
    kst.load(K + base, ldi, dh, 0, n, tid);
    auto load_ds = [&](int kc, f32x4 (&d)[NKT]) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int k0 = kc + 16 * kt + 4 * g;
            if (vds) d[kt] = *reinterpret_cast<const f32x4 *>(dsrow + (k0 + 3 < L ? k0 : 0));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) d[kt][r] = dsrow[k0 + r < L ? k0 + r : 0];
            }
        }
    };
    f32x4 dsn[NKT];
    load_ds(0, dsn);
    for (int kc = 0; kc < n; kc += KC) {
        __syncthreads();
        kst.store(Ks, dh, kc, n, tid);
        __syncthreads();
        f32x4 ds[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[kt][r] = (rok && kc + 16 * kt + 4 * g + r < n) ? dsn[kt][r] : 0.0f;
        if (kc + KC < n) { kst.load(K + base, ldi, dh, kc + KC, n, tid); load_ds(kc + KC, dsn); }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *krow = Ks + (size_t)(16 * kt + 4 * g + r) * ld + j;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = mfma4(krow[16 * dt], ds[kt][r], dq[dt]);
            }
        }
    }
    if (!rok) return;
    const bool vec = ((dh & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(dQ + base) & 15) == 0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        float *dst = dQ + base + (size_t)row * ldi + d;
        if (vec) { if (d < dh) *reinterpret_cast<f32x4 *>(dst) = dq[dt]; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (d + e < dh) dst[e] = dq[dt][e];
        }
    }
}

// ============================================================================================ backward: dK, dV
// STORE_DS: the variant that also hands the scaled dS to mhsa_bwd_dq_ds_kernel; compiled for two workgroups per CU (the transpose of the
// dS tiles does not fit the 168 registers of three: 49 spills, +280 us per launch at config 5)
template <int DT, int NW, bool STORE_DS>
__global__ void __launch_bounds__(NW * 64, STORE_DS ? 2 : PTR_ATTN_MINBLK_BWD)
mhsa_bwd_dkv_kernel(const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V,
                    const float *__restrict__ dO, const float *__restrict__ LSE, const float *__restrict__ Dv,
                    const int32_t *__restrict__ lens, AttnArgs a, float *__restrict__ dK, float *__restrict__ dV,
                    float *__restrict__ dS_ws /* nullable: [B*H][L][L] scaled dS for mhsa_bwd_dq_ds_kernel */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ld = attn_ld(DT), KPB = 16 * NW, NT = NW * 64;
    float *Qs = smem, *Gs = Qs + (size_t)kRC * ld;
    float *lse_s = Gs + (size_t)kRC * ld, *D_s = lse_s + kRC;
    float *ds_pad = D_s + kRC;                            // [NW][16][kDsPadLd] wave-private transpose pads of the dS tiles
    const int L = a.L, F = a.F, dh = a.dh, ldi = a.ld;
    const int nkb = (L + KPB - 1) / KPB;
    const int lid = xcd_major_block_id();
    const int kb = lid % nkb, bh = lid / nkb, b = bh / a.H, h = bh - b * a.H;
    int n = lens ? lens[b] : L;
    n = n < 0 ? 0 : (n > L ? L : n);
    const size_t base = (size_t)b * L * ldi + (size_t)h * dh, obase = (size_t)b * L * F + (size_t)h * dh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int key0 = kb * KPB, wkey = wave * 16;
    const int key = key0 + wkey + j;
    const int dh4 = (dh + 3) & ~3, nb = dh4 >> 4, rem = (dh4 & 15) >> 2;
    const uint32_t thr = drop_thr(a.p_drop);
    const float keep_inv = thr != 0 ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const bool live = key0 < n;                            // a block of padded keys only writes zeros

    RowStage<kRC, ld, NT> qst, gst;
    float lse_pf = 0.0f, d_pf = 0.0f;
    auto prefetch = [&](int rc) {
        qst.load(Q + base, ldi, dh, rc, L, tid);
        gst.load(dO + obase, F, dh, rc, L, tid);
        if (tid < kRC) {
            const int row = rc + tid, rcl = row < L ? row : L - 1;     // raw loads; rows >= L are masked where they are used
            lse_pf = LSE[(size_t)bh * L + rcl];
            d_pf = Dv[(size_t)bh * L + rcl];
        }
    };
    if (live) prefetch(0);
    OperandRegs<DT> kr[1], vr[1];                          // this wave's 16 keys as B operands, for the whole kernel
    {
        const bool vk = ((dh & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(K + base) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(V + base) & 15) == 0);
        const bool kok = key < n;
        const size_t ko = (size_t)(key < L ? key : L - 1) * ldi;
        kr[0].load_global(K + base + ko, kok, dh, nb, rem, g, vk);
        vr[0].load_global(V + base + ko, kok, dh, nb, rem, g, vk);
    }
    for (int rc = 0; live && rc < L; rc += kRC) {
        __syncthreads();
        qst.store(Qs, dh, rc, L, tid);
        gst.store(Gs, dh, rc, L, tid);
        if (tid < kRC) { lse_s[tid] = lse_pf; D_s[tid] = d_pf; }
        __syncthreads();
        if (rc + kRC < L) prefetch(rc + kRC);
        // S[row][key], dP[row][key] for all row tiles of the chunk: lane (j, g), reg r = row 16*rt + 4*g + r, key j
        f32x4 s4[1][kRC / 16], dp4[1][kRC / 16];
        multi_dot<DT, kRC / 16, 1>(Qs, 0, ld, nb, rem, j, g, kr, s4);
        multi_dot<DT, kRC / 16, 1>(Gs, 0, ld, nb, rem, j, g, vr, dp4);
#pragma unroll
        for (int rt = 0; rt < kRC / 16; ++rt) {
            const f32x4 s = s4[0][rt], dp = dp4[0][rt];
            f32x4 pd, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 16 * rt + 4 * g + r, row = rc + lr;
                const float pr = (key < n && row < L) ? __expf(s[r] * a.inv_scale - lse_s[lr]) : 0.0f;
                float keep = keep_inv;
                if (thr != 0) keep = drop_keep1(a.seed_lo, a.seed_hi, a.site, bh * L + row, key, thr) ? keep_inv : 0.0f;
                pd[r] = pr * keep;
                ds[r] = pr * (dp[r] * keep - D_s[lr]) * a.inv_scale;
            }
            // r3: hand dS to the dQ kernel instead of letting it recompute S and dP (two of its three GEMM units): 4 L^2 bytes per
            // (query, head) through HBM — 0.54 GB per layer at config 5, cheap against 157 TFLOP/s of fp32 MFMA.  The tile sits in the
            // C layout (lane = key, registers = 4 rows); it is transposed through a wave-private LDS pad so that every lane stores ONE
            // float4 of 4 consecutive keys of its row (16 rows x 64 contiguous bytes per tile; scalar stores took 45 % longer than the
            // whole rest of the kernel).
            if constexpr (STORE_DS) {
                float *T = ds_pad + wave * (16 * kDsPadLd);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(4 * g + r) * kDsPadLd + j] = ds[r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f32x4 t4 = *reinterpret_cast<const f32x4 *>(T + j * kDsPadLd + 4 * g);       // row 16 rt + j, keys 4 g .. 4 g + 3 of the tile
                const int srow = rc + 16 * rt + j, skey = key0 + wkey + 4 * g;
                if (srow < L) {
                    float *dst = dS_ws + ((size_t)bh * L + srow) * L + skey;
                    if (skey + 3 < L && ((L & 3) == 0)) *reinterpret_cast<f32x4 *>(dst) = t4;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (skey + e < L) dst[e] = t4[e];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *grow = Gs + (size_t)(16 * rt + 4 * g + r) * ld + j;
                const float *qrow = Qs + (size_t)(16 * rt + 4 * g + r) * ld + j;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = mfma4(grow[16 * dt], pd[r], dv[dt]);        // dV^T[d][key] += dO^T[d][row] * Pdrop[row][key]
                    dk[dt] = mfma4(qrow[16 * dt], ds[r], dk[dt]);        // dK^T[d][key] += Q^T[d][row]  * dS[row][key]
                }
            }
        }
    }
    if (key >= L) return;
    const bool vec = ((dh & 3) == 0) && ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(dK + base) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dV + base) & 15) == 0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        float *pk = dK + base + (size_t)key * ldi + d, *pv = dV + base + (size_t)key * ldi + d;
        if (vec) { if (d < dh) { *reinterpret_cast<f32x4 *>(pk) = dk[dt]; *reinterpret_cast<f32x4 *>(pv) = dv[dt]; } }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (d + e < dh) { pk[e] = dk[dt][e]; pv[e] = dv[dt][e]; }
        }
    }
}

// Test helper: the keep mask (1 / 0) of the attention dropout, [B][H][L][L].
__global__ void __launch_bounds__(256) attn_mask_kernel(AttnArgs a, float *__restrict__ out) {
    const size_t tot = (size_t)a.B * a.H * a.L * a.L;
    const uint32_t thr = drop_thr(a.p_drop);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const int key = (int)(i % a.L);
        const size_t grow = i / a.L;
        out[i] = (thr == 0 || drop_keep1(a.seed_lo, a.seed_hi, a.site, (int)grow, key, thr)) ? 1.0f : 0.0f;
    }
}

// ============================================================================================ LayerNorm (list_ranker.py:152-174)
// y = a_2 * (x - mean) / (std + eps) + b_2, std UNBIASED (torch.Tensor.std default).  One wavefront per row, lanes along
// the feature axis (coalesced 256-byte segments); HBM-bound streaming kernels.  stats[row] = {mean, 1/(std + eps), std}.
// NI = ceil(F / 64) when the row fits in NI registers per lane (read once), 0 = generic re-reading loops.
template <int NI>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float *__restrict__ X, const float *__restrict__ a2, const float *__restrict__ b2, size_t R, int F,
                     float eps, float *__restrict__ Y, float *__restrict__ stats) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t r = (size_t)blockIdx.x * 4 + wv; r < R; r += (size_t)gridDim.x * 4) {
        const float *x = X + r * F;
        if constexpr (NI > 0) {
            float xv[NI];
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int d = lane + 64 * i;
                xv[i] = x[d < F ? d : F - 1] * (d < F ? 1.0f : 0.0f);
                s += xv[i];
            }
            const float mean = wave_sum(s) / (float)F;
            float v = 0.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i) { const float c = (lane + 64 * i < F) ? xv[i] - mean : 0.0f; v += c * c; }
            const float sd = sqrtf(wave_sum(v) / (float)(F - 1));
            const float rinv = 1.0f / (sd + eps);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int d = lane + 64 * i;
                if (d < F) Y[r * F + d] = a2[d] * (xv[i] - mean) * rinv + b2[d];
            }
            if (lane == 0) { stats[3 * r] = mean; stats[3 * r + 1] = rinv; stats[3 * r + 2] = sd; }
        } else {
            float s = 0.0f;
            for (int d = lane; d < F; d += 64) s += x[d];
            const float mean = wave_sum(s) / (float)F;
            float v = 0.0f;
            for (int d = lane; d < F; d += 64) { const float c = x[d] - mean; v += c * c; }
            const float sd = sqrtf(wave_sum(v) / (float)(F - 1));
            const float rinv = 1.0f / (sd + eps);
            for (int d = lane; d < F; d += 64) Y[r * F + d] = a2[d] * (x[d] - mean) * rinv + b2[d];
            if (lane == 0) { stats[3 * r] = mean; stats[3 * r + 1] = rinv; stats[3 * r + 2] = sd; }
        }
    }
}

// dX, and per-block partial sums of da_2 / db_2 in part[gridDim.x][2*F] (reduced by layernorm_reduce_kernel).
template <int NI>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const float *__restrict__ X, const float *__restrict__ a2, const float *__restrict__ dY,
                     const float *__restrict__ stats, size_t R, int F, float *__restrict__ dX, float *__restrict__ part) {
    extern __shared__ float red[];                         // [4 waves][2*F]: every lane only touches its own columns
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *mine = red + (size_t)wv * 2 * F;
    for (int d = lane; d < 2 * F; d += 64) mine[d] = 0.0f;
    float pa[NI > 0 ? NI : 1], pb[NI > 0 ? NI : 1], av[NI > 0 ? NI : 1];
    if constexpr (NI > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; pa[i] = 0.0f; pb[i] = 0.0f; av[i] = d < F ? a2[d] : 0.0f; }
    }
    for (size_t r = (size_t)blockIdx.x * 4 + wv; r < R; r += (size_t)gridDim.x * 4) {
        const float *x = X + r * F, *dy = dY + r * F;
        const float mean = stats[3 * r], rinv = stats[3 * r + 1], sd = stats[3 * r + 2];
        float sg = 0.0f, sgc = 0.0f;
        if constexpr (NI > 0) {
            float cv[NI], gv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int d = lane + 64 * i, dc = d < F ? d : F - 1;
                const float ok = d < F ? 1.0f : 0.0f;
                cv[i] = (x[dc] - mean) * ok;
                gv[i] = dy[dc] * ok;
                const float gg = gv[i] * av[i];
                sg += gg; sgc += gg * cv[i];
                pa[i] += gv[i] * cv[i] * rinv;
                pb[i] += gv[i];
            }
            sg = wave_sum(sg); sgc = wave_sum(sgc);
            const float gbar = sg / (float)F;
            const float k2 = sd > 0.0f ? rinv * rinv * sgc / (sd * (float)(F - 1)) : 0.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int d = lane + 64 * i;
                if (d < F) dX[r * F + d] = rinv * (gv[i] * av[i] - gbar) - cv[i] * k2;
            }
        } else {
            for (int d = lane; d < F; d += 64) {
                const float c = x[d] - mean, gy = dy[d], gg = gy * a2[d];
                sg += gg; sgc += gg * c;
                mine[d] += gy * c * rinv;                      // d a_2
                mine[F + d] += gy;                             // d b_2
            }
            sg = wave_sum(sg); sgc = wave_sum(sgc);
            const float gbar = sg / (float)F;
            // d(1/(std+eps))/dx_k = -rinv^2 * (x_k - mean) / (std * (F-1));  a constant row (std = 0) has no defined
            // derivative (the reference's autograd returns NaN there): its second term is dropped
            const float k2 = sd > 0.0f ? rinv * rinv * sgc / (sd * (float)(F - 1)) : 0.0f;
            for (int d = lane; d < F; d += 64) dX[r * F + d] = rinv * (dy[d] * a2[d] - gbar) - (x[d] - mean) * k2;
        }
    }
    if constexpr (NI > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < F) { mine[d] = pa[i]; mine[F + d] = pb[i]; } }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 2 * F; d += 256)
        part[(size_t)blockIdx.x * 2 * F + d] = (red[d] + red[(size_t)2 * F + d]) + (red[(size_t)4 * F + d] + red[(size_t)6 * F + d]);
}

// Fixed-order reduction of the per-block partials: 16 columns x 16 partial lanes per workgroup (lane sl sums the partials sl, sl + 16, ...
// four at a time; the 16 lane sums are added in lane order).
__global__ void __launch_bounds__(256)
layernorm_reduce_kernel(const float *__restrict__ part, int nblk, int F, float *__restrict__ da2, float *__restrict__ db2) {
    constexpr int CL = 16, BL = 16;
    __shared__ float red[BL][CL + 1];
    const int cl = threadIdx.x & (CL - 1), sl = threadIdx.x / CL;
    const int d = blockIdx.x * CL + cl;
    const bool on = d < 2 * F;
    auto at = [&](int k) { return k < nblk ? part[(size_t)k * 2 * F + d] : 0.0f; };
    float s = 0.0f;
    if (on)
        for (int k = sl; k < nblk; k += 4 * BL) s += (at(k) + at(k + BL)) + (at(k + 2 * BL) + at(k + 3 * BL));
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && on) {
        float t = 0.0f;
        for (int k = 0; k < BL; ++k) t += red[k][cl];
        if (d < F) da2[d] = t; else db2[d - F] = t;
    }
}

constexpr int kLnBlocks = 1024;

template <class Fn> inline int dispatch_dt(int DT, Fn &&f) {
    switch (DT) {
        case 1: return f.template operator()<1>();
        case 2: return f.template operator()<2>();
        case 3: return f.template operator()<3>();
        case 4: return f.template operator()<4>();
        case 5: return f.template operator()<5>();
        case 6: return f.template operator()<6>();
        case 7: return f.template operator()<7>();
        default: return f.template operator()<8>();
    }
}

static int attn_args(const char *who, int B, int L, int F, int H, int ld, float p_drop, uint64_t seed, int site, AttnArgs &a) {
    if (ld < F) { set_error("%s: row stride %d < F = %d", who, ld, F); return PTR_ERR_INVALID_ARG; }
    if (B < 0 || L <= 0 || F <= 0 || H <= 0 || F % H != 0) { set_error("%s: bad shape B=%d L=%d F=%d heads=%d", who, B, L, F, H); return PTR_ERR_INVALID_ARG; }
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) { set_error("%s: p_drop must be in [0,1)", who); return PTR_ERR_INVALID_ARG; }
    const int dh = F / H;
    if (dh > PTR_MHSA_MAX_HEAD_DIM) { set_error("%s: head dimension %d exceeds PTR_MHSA_MAX_HEAD_DIM=%d", who, dh, PTR_MHSA_MAX_HEAD_DIM); return PTR_ERR_UNSUPPORTED; }
    if ((size_t)B * H * L >= (1u << 31)) { set_error("%s: B*H*L too large", who); return PTR_ERR_UNSUPPORTED; }
    a = AttnArgs{B, L, H, dh, F, ld, 1.0f / sqrtf((float)dh), p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site};
    return 0;
}

}  // namespace ptr

// Waves per workgroup: 4 (default) lets two independent workgroups share a CU, so one group's barrier / staging phases are
// covered by the other's MFMA work; 8 halves the K/V re-staging traffic.  PTR_ATTN_WAVES=8 selects the latter.
static int attn_waves() {
    static int nw = [] { const char *e = getenv("PTR_ATTN_WAVES"); return (e && atoi(e) == 8) ? 8 : 4; }();
    return nw;
}

extern "C" int ptr_mhsa_forward(const float *Q, const float *K, const float *V, int ld_qkv, const int32_t *lens, int B, int L, int F,
                                int n_heads, float p_drop, uint64_t seed, int site, float *O, float *lse, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mhsa_forward";
    AttnArgs a;
    if (int rc = attn_args(who, B, L, F, n_heads, ld_qkv, p_drop, seed, site, a)) return rc;
    if (B == 0) return 0;
    if (!Q || !K || !V || !O || !lse) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    const int DT = (a.dh + 15) / 16;
    return dispatch_dt(DT, [&]<int D>() -> int {
        auto launch = [&]<int RT, int NW>() -> int {
            constexpr int RPB = 16 * RT * NW;
            auto kern = mhsa_fwd_kernel<D, RT, NW>;
            const size_t lds = (size_t)2 * 8 * NW * attn_ld(D) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            const int nrb = (L + RPB - 1) / RPB;
            hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * n_heads * nrb)), dim3(NW * 64), lds, st, Q, K, V, lens, a, O, lse);
            return check_hip(hipGetLastError(), who);
        };
        // two row tiles per wave (every K / V operand read feeds two MFMAs) when the rows exist and the registers allow it
        static const int rt1 = [] { const char *e = getenv("PTR_ATTN_RT1"); return e ? atoi(e) : 0; }();    // measurements
        if constexpr (D <= 5) {
            if (L > 64 && !rt1) return attn_waves() == 8 && L > 128 ? launch.template operator()<2, 8>() : launch.template operator()<2, 4>();
        }
        return attn_waves() == 8 && L > 64 ? launch.template operator()<1, 8>() : launch.template operator()<1, 4>();
    });
}

extern "C" int ptr_mhsa_backward(const float *Q, const float *K, const float *V, int ld_qkv, const float *O, const float *dO,
                                 const float *lse, const int32_t *lens, int B, int L, int F, int n_heads, float p_drop, uint64_t seed,
                                 int site, float *dvec, float *dQ, float *dK, float *dV, float *ds_ws, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mhsa_backward";
    AttnArgs a;
    if (int rc = attn_args(who, B, L, F, n_heads, ld_qkv, p_drop, seed, site, a)) return rc;
    if (B == 0) return 0;
    if (!Q || !K || !V || !O || !dO || !lse || !dvec || !dQ || !dK || !dV) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    const size_t nrows = (size_t)B * n_heads * L;
    const int rblocks = (int)((nrows + 15) / 16 < 4096 ? (nrows + 15) / 16 : 4096);
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3(rblocks), dim3(256), 0, st, O, dO, a, dvec);
    if (int rc = check_hip(hipGetLastError(), who)) return rc;
    const int DT = (a.dh + 15) / 16;
    return dispatch_dt(DT, [&]<int D>() -> int {
        auto launch_dq = [&]<int NW>() -> int {
            constexpr int RPB = 16 * NW;
            auto kern = mhsa_bwd_dq_kernel<D, NW>;
            const size_t lds = (size_t)2 * 8 * NW * attn_ld(D) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            const int nrb = (L + RPB - 1) / RPB;
            hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * n_heads * nrb)), dim3(NW * 64), lds, st, Q, K, V, dO, lse, dvec, lens, a, dQ);
            return check_hip(hipGetLastError(), who);
        };
        auto launch_dkv = [&]<int NW>() -> int {
            constexpr int KPB = 16 * NW;
            const size_t lds = ((size_t)2 * kRC * attn_ld(D) + 2 * kRC + (size_t)NW * 16 * kDsPadLd) * sizeof(float);
            const int nkb = (L + KPB - 1) / KPB;
            auto go = [&](auto kern) -> int {
                if (int e = allow_lds(kern, lds)) return e;
                hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * n_heads * nkb)), dim3(NW * 64), lds, st, Q, K, V, dO, lse, dvec, lens, a, dK, dV, ds_ws);
                return check_hip(hipGetLastError(), who);
            };
            return ds_ws ? go(mhsa_bwd_dkv_kernel<D, NW, true>) : go(mhsa_bwd_dkv_kernel<D, NW, false>);
        };
        auto launch_dq_ds = [&]<int NW>() -> int {
            constexpr int RPB = 16 * NW;
            auto kern = mhsa_bwd_dq_ds_kernel<D, NW>;
            const size_t lds = (size_t)8 * NW * attn_ld(D) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            const int nrb = (L + RPB - 1) / RPB;
            hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * n_heads * nrb)), dim3(NW * 64), lds, st, K, ds_ws, lens, a, dQ);
            return check_hip(hipGetLastError(), who);
        };
        const bool w8 = attn_waves() == 8 && L > 64 && D <= 6;     // the 8-wave variants of the widest heads would spill
        const bool kv8 = L > 64 && (D >= 7 || attn_waves() == 8);   // 4-wave dK/dV variants of the widest heads spill more
        if (ds_ws) {   // dK / dV first (it writes dS), then dQ as ONE GEMM unit from the stored dS
            if (int rc = kv8 ? launch_dkv.template operator()<8>() : launch_dkv.template operator()<4>()) return rc;
            return w8 ? launch_dq_ds.template operator()<8>() : launch_dq_ds.template operator()<4>();
        }
        int rc = w8 ? launch_dq.template operator()<8>() : launch_dq.template operator()<4>();
        if (rc) return rc;
        return kv8 ? launch_dkv.template operator()<8>() : launch_dkv.template operator()<4>();
    });
}

extern "C" int ptr_mhsa_dropout_mask(int B, int L, int n_heads, float p_drop, uint64_t seed, int site, float *out, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mhsa_dropout_mask";
    AttnArgs a;
    if (int rc = attn_args(who, B, L, n_heads, n_heads, n_heads, p_drop, seed, site, a)) return rc;
    if (B == 0) return 0;
    if (!out) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(attn_mask_kernel, dim3(2048), dim3(256), 0, as_stream(stream), a, out);
    return check_hip(hipGetLastError(), who);
}

extern "C" int ptr_layernorm_forward(const float *X, const float *a2, const float *b2, int64_t R, int F, float eps, float *Y,
                                     float *stats, void *stream) {
    using namespace ptr;
    const char *who = "ptr_layernorm_forward";
    if (R < 0 || F < 2) { set_error("%s: bad shape R=%lld F=%d", who, (long long)R, F); return PTR_ERR_INVALID_ARG; }
    if (R == 0) return 0;
    if (!X || !a2 || !b2 || !Y || !stats) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    const int blocks = (int)((R + 3) / 4 < 8192 ? (R + 3) / 4 : 8192);
    const int ni = (F + 63) / 64;
    auto launch = [&]<int NI>() {
        hipLaunchKernelGGL(layernorm_fwd_kernel<NI>, dim3(blocks), dim3(256), 0, as_stream(stream), X, a2, b2, (size_t)R, F, eps, Y, stats);
    };
    if (ni == 1) launch.template operator()<1>(); else if (ni == 2) launch.template operator()<2>();
    else if (ni == 3) launch.template operator()<3>(); else if (ni == 4) launch.template operator()<4>();
    else launch.template operator()<0>();
    return check_hip(hipGetLastError(), who);
}

extern "C" size_t ptr_layernorm_backward_ws_floats(int F) { return (size_t)ptr::kLnBlocks * 2 * (size_t)(F > 0 ? F : 0); }

extern "C" int ptr_layernorm_backward(const float *X, const float *a2, const float *dY, const float *stats, int64_t R, int F, float *ws,
                                      float *dX, float *da2, float *db2, void *stream) {
    using namespace ptr;
    const char *who = "ptr_layernorm_backward";
    if (R < 0 || F < 2) { set_error("%s: bad shape R=%lld F=%d", who, (long long)R, F); return PTR_ERR_INVALID_ARG; }
    if (!a2 || !da2 || !db2 || !ws || (R > 0 && (!X || !dY || !stats || !dX))) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    const size_t lds = (size_t)4 * 2 * F * sizeof(float);
    if (lds > 160 * 1024) { set_error("%s: F=%d too wide", who, F); return PTR_ERR_UNSUPPORTED; }
    const int ni = (F + 63) / 64;
    hipStream_t st = as_stream(stream);
    int blocks = (int)((R + 3) / 4 < kLnBlocks ? (R + 3) / 4 : kLnBlocks);
    if (blocks < 1) blocks = 1;
    auto launch = [&]<int NI>() -> int {
        if (int e = allow_lds(layernorm_bwd_kernel<NI>, lds)) return e;
        hipLaunchKernelGGL(layernorm_bwd_kernel<NI>, dim3(blocks), dim3(256), lds, st, X, a2, dY, stats, (size_t)R, F, dX, ws);
        return check_hip(hipGetLastError(), who);
    };
    int rc0;
    if (ni == 1) rc0 = launch.template operator()<1>(); else if (ni == 2) rc0 = launch.template operator()<2>();
    else if (ni == 3) rc0 = launch.template operator()<3>(); else if (ni == 4) rc0 = launch.template operator()<4>();
    else rc0 = launch.template operator()<0>();
    if (rc0) return rc0;
    hipLaunchKernelGGL(layernorm_reduce_kernel, dim3((2 * F + 15) / 16), dim3(256), 0, st, ws, blocks, F, da2, db2);
    return check_hip(hipGetLastError(), who);
}
