// Fused fp32-MFMA pointwise MLP scorer (the reference's `pointsf`): forward, backward and a flat Adam step.
//
// Reference: ptranking/base/point_ranker.py:30-55 (ff_dims = [F] + [100]*num_layers + [1], forward = Sequential(X).view(-1, L)),
//            ptranking/base/utils.py:288-356 (get_stacked_FFNet with AF='R', BN=False, apply_tl_af=False:
//            (Dropout(0.1) -> Linear -> ReLU) x num_layers -> Linear), ptranking/base/ranker.py:512-525 (Adam, weight_decay 1e-3).
// Every document is scored independently, so the batch is a matrix of R = B*L rows x F features.
//
// Parameters live in ONE flat fp32 buffer in PyTorch's own order and layouts (Linear.weight is [out][in]):
//   W1[100][F] b1[100] | W2[100][100] b2[100] | ... | w_out[100] b_out[1]
// so the optimiser, the data-parallel all-reduce and checkpoints see a single tensor.
//
// MFMA formulation (v_mfma_f32_16x16x4_f32: exact fp32, 256 FLOP/clk/CU = the fp32 vector peak, 157 TFLOP/s):
//   forward, "transposed world":  Z^T[feature][row] = W[feature][k] * A^T[k][row].  A 16x16 output tile leaves lane (j = l&15,
//   g = l>>4) holding row j's features 16*mt + 4*g + {0..3} — which IS the B-operand layout of the next layer (k may be
//   visited in any order as long as A and B agree), so activations stay in registers through all layers; LDS only holds
//   the weights (A operands, read as one ds_read_b128 per 4 k-steps straight from the [out][in] layout).
//   backward dZ, same world with W^T in LDS:  dA^T[k][row] = W^T[k][out] * dZ^T[out][row].
//   backward dW, "row-contraction world":     dW[out][in] = sum_rows dZ[row][out] * A[row][in]  streams dZ and A from HBM with
//   rows as the MFMA k index (lane j = feature, g = row), accumulating the whole 100 x K gradient in registers per wave.
// Dropout masks come from a counter-based hash of (seed, site, row, feature/4) and are recomputed, never stored.
//
// HBM traffic per row (F = 136, 3 hidden layers, training): forward reads 4F, writes 3*400 + 4; dZ reads 3*400 + 4, writes
// 3*400; dW reads 4F + 5*400.  MFMA work per row: 2*(100F + 2*100*100 + 100) flop forward, about twice that backward.
#include "ptr_mlp.h"

namespace ptr {

#ifndef DW_U
#define DW_U 4
#endif
__host__ __device__ inline int ld_w1(int F) { return (F + 3) / 4 * 4 + 4; }   // LDS leading dimension of W1 (bank spread)


// Stage a [rows_valid][cols_valid] row-major matrix into LDS as [kHP][ld], zero padded.  transpose: dst[c][r] = src[r][c].
// The source is walked linearly (coalesced) in batches of U INDEPENDENT loads — a plain `for (...) dst[i] = src[...]` loop issues
// one load per iteration and waits for it (75 dependent L2 round trips per thread = 25 us of prologue at F = 136); the zero padding
// of the [kHP][ld] tile is written separately (disjoint elements, no barrier needed in between).
// tail (forward only, rows == kH = 100): how the output features 96..99 are laid out on the LDS rows / elements 96..111
//   kTailNone       natural: 96..99, zeros behind
//   kTailSpread     element 96 + 4 i holds feature 96 + i, zeros elsewhere (w_out: lane group g reads {w[96 + g], 0, 0, 0})
//   kTailReplicate  row 96 + r holds feature 96 + (r & 3) (weight matrices): with (16 * 6 + j) row addressing lane j reads feature
//                   96 + (j & 3) — the A operand of v_mfma_f32_4x4x1_16b_f32, which multiplies M tile 6 (4 real features) as 16 blocks of
//                   4 features x 4 documents x 1 k instead of a 16 x 16 x 4 tile that is three quarters padding
constexpr int kTailNone = 0, kTailSpread = 1, kTailReplicate = 2;
__device__ __forceinline__ void stage_matrix(float *dst, int ld, const float *src, int rows, int cols, bool transpose, int tid, int nthr,
                                             int tail = kTailNone) {
    constexpr int U = 8;
    const int n = rows * cols;
    const bool vec = ((cols & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const int reps = tail == kTailReplicate ? 4 : 1;
    if (vec) {
        const int n4 = n >> 2;
        for (int base = tid; base < n4; base += U * nthr) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = base + u * nthr;
                v[u] = *reinterpret_cast<const f32x4 *>(src + 4 * (size_t)(i4 < n4 ? i4 : n4 - 1));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = base + u * nthr;
                if (i4 < n4) {
                    const int idx = 4 * i4, r = idx / cols, c = idx - r * cols;
                    if (!transpose) {
                        for (int rep = 0; rep < (r >= 96 ? reps : 1); ++rep) *reinterpret_cast<f32x4 *>(dst + (size_t)(r + 4 * rep) * ld + c) = v[u];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(size_t)(c + e) * ld + r] = v[u][e];
                    }
                }
            }
        }
    } else {
        for (int base = tid; base < n; base += U * nthr) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int idx = base + u * nthr; v[u] = src[idx < n ? idx : n - 1]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * nthr;
                if (idx < n) {
                    const int r = idx / cols, c = idx - r * cols;
                    if (transpose) dst[(size_t)c * ld + r] = v[u];
                    else for (int rep = 0; rep < (r >= 96 ? reps : 1); ++rep) dst[(size_t)(r + 4 * rep) * ld + c] = v[u];
                }
            }
        }
    }
    const int vr = transpose ? cols : rows, vc = transpose ? rows : cols;
    for (int idx = tid; idx < kHP * ld; idx += nthr) {
        const int r = idx / ld, c = idx - r * ld;
        const bool real_row = (tail == kTailReplicate && r >= 96) ? 96 + ((r - 96) & 3) < vr : r < vr;
        if (!real_row || c >= vc) dst[idx] = 0.0f;
    }
}
__device__ __forceinline__ void stage_vector(float *dst, const float *src, int n, int tid, int nthr, int tail = kTailNone) {
    for (int i = tid; i < kHP; i += nthr) {
        if (tail != kTailSpread || i < 96) dst[i] = i < n ? src[i] : 0.0f;
        else dst[i] = (((i - 96) & 3) == 0 && 96 + ((i - 96) >> 2) < n) ? src[96 + ((i - 96) >> 2)] : 0.0f;
    }
}

// =================================================================================================== forward
// LDS: W1s [kHP][ld1] | Wh (NL-1) x [kHP][kH] | B NL x [kHP] | Wo [kHP] | bo + pad [16]
// First-layer weights that do not fit next to the hidden weights (F above ~157) stay in global memory (L2-resident) and reach the MFMAs
//   w1_mode 2 (slab): through two LDS slabs of 48 k (three super-steps) that the workgroup stages cooperatively one slab ahead — every
//                     W1 element is fetched from L2 once per workgroup and group of tiles, and the fragment reads have LDS latency;
//   w1_mode 1 (stream): by per-wave global loads, when not even the slabs fit (four hidden layers).
constexpr int kSlabK = 48, kSlabLd = kSlabK + 4;      // 52 = 4 * 13: 8 consecutive rows cover the 32 banks with their float4s
__host__ __device__ inline size_t fwd_lds_floats(int F, int NL, int w1_mode) {
    const size_t w1 = w1_mode == 0 ? (size_t)kHP * ld_w1(F) : (w1_mode == 2 ? (size_t)2 * kHP * kSlabLd : 0);
    return w1 + (size_t)(NL - 1) * kHP * kH + (size_t)NL * kHP + kHP + 16;
}
__host__ __device__ inline int fwd_w1_mode(int F, int NL) {
    if (fwd_lds_floats(F, NL, 0) * sizeof(float) <= 160 * 1024) return 0;
    return fwd_lds_floats(F, NL, 2) * sizeof(float) <= 160 * 1024 ? 2 : 1;
}

#ifdef PTR_FWD_TRACE   // experiment builds: shader-clock stamps of workgroup 0's first tiles behind the predictions (preds is over-allocated)
#define FWD_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && ntile_done < 8) { unsigned long long *tr_ = reinterpret_cast<unsigned long long *>(preds + ((R + 3) & ~3) + 4) + (ntile_done * 16 + wave) * 8; tr_[(i)] = clock64(); if ((i) == 0) tr_[4] = wall_clock64(); if ((i) == 3) tr_[5] = wall_clock64(); } } while (0)   /* slots 4/5: 100 MHz real-time counter at tile start / end */
#else
#define FWD_STAMP(i) do { } while (0)
#endif

// TQ: K tail of layer 1 (see load_raw): 0 = zero-padded last super-step; 1 / 2 = F mod 16 of 4 / 8 features (VEC, W1 in LDS) taken in
// 1 / 2 MFMAs per output tile
template <int RT, int NTHR, bool TRAIN, bool VEC, int W1G, int TQ>
__global__ void __launch_bounds__(NTHR)
mlp_fwd_kernel(const float *__restrict__ X, const float *__restrict__ P, MlpArgs a, float *__restrict__ preds,
               float *__restrict__ acts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = a.F, NL = a.NL, R = a.R, ld1 = ld_w1(F), R16 = act_row_tiles(a.R) * 16;
    float *W1s = smem;
    float *Wh = W1s + (W1G == 0 ? (size_t)kHP * ld1 : (W1G == 2 ? (size_t)2 * kHP * kSlabLd : 0));
    float *Bs = Wh + (size_t)(NL - 1) * kHP * kH;
    float *Wo = Bs + (size_t)NL * kHP;
    const int tid = threadIdx.x, nthr = blockDim.x;
    // output features 96..99: replicated over rows 96..111 of the weight matrices (4x4 MFMA blocks of M tile 6), spread over elements
    // 96 / 100 / 104 / 108 of w_out (lane group g ends up with feature 96 + g), natural in the biases
    if constexpr (W1G == 0) stage_matrix(W1s, ld1, P + off_W(0, F), kH, F, false, tid, nthr, kTailReplicate);
    for (int l = 1; l < NL; ++l) stage_matrix(Wh + (size_t)(l - 1) * kHP * kH, kH, P + off_W(l, F), kH, kH, false, tid, nthr, kTailReplicate);
    for (int l = 0; l < NL; ++l) stage_vector(Bs + (size_t)l * kHP, P + off_b(l, F), kH, tid, nthr);
    stage_vector(Wo, P + off_wout(NL, F), kH, tid, nthr, kTailSpread);
    if (tid < 16) Wo[kHP + tid] = tid == 0 ? P[off_wout(NL, F) + kH] : 0.0f;
    __syncthreads();
    const float b_out = Wo[kHP];

    const int lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wpb = nthr >> 6, wave = tid >> 6;
    const int rows_per_tile = 16 * RT;
    const int ntiles = (R + rows_per_tile - 1) / rows_per_tile;
    const uint32_t thr = drop_thr(a.p_drop);
    const float scale = TRAIN ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int nS1 = (F + 15) >> 4;
    // K tail of layer 1 (see load_raw): rem = F mod 16 of 4 or 8 features with W1 in LDS and float4 X loads; other shapes keep the
    // zero-padded last super-step
    const int nSf = F >> 4;
    constexpr int tq = TQ;

    int ntile_done = 0;
    (void)ntile_done;
    // First X super-step of a tile, issued one tile AHEAD (while the previous tile runs its last hidden layer): the loads are then
    // older than that tile's activation stores, so waiting for them does not wait for the stores' acknowledgements (vmcnt counts
    // in order), and their HBM latency is off the tile's critical path.  Past the end: a valid address, never consumed.
    auto prefetch_first = [&](int t, f32x4 (&xb)[RT]) {
        const int tt = t < ntiles ? t : ntiles - 1;
        const int k0 = (tq > 0 && nSf == 0) ? (tq == 2 ? 4 * (g >> 1) : 0) : 4 * g;      // F < 16: the first super-step is the K tail
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = tt * rows_per_tile + 16 * rt + j;
            const float *xr = X + (size_t)(r < R ? r : R - 1) * F;
            if constexpr (VEC) {
                xb[rt] = *reinterpret_cast<const f32x4 *>(xr + (k0 < F ? k0 : 0));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[rt][c] = xr[k0 + c < F ? k0 + c : 0];
            }
        }
    };
    f32x4 xpre[RT];
#ifndef PTR_FWD_STATIC_TILES
    constexpr bool kQueue = W1G != 2;
#else
    constexpr bool kQueue = false;
#endif
    // dynamic tile queue per workgroup: the two waves of a SIMD do not progress at the same rate (the older one wins the issue
    // arbitration: 90 K vs 145 K cycles per tile measured) — with a static split the faster half idles at the end.
    // Slab mode (W1G == 2): the waves of a workgroup share the W1 slabs and walk their tiles in LOCKSTEP (one barrier per slab), so a
    // group of wpb consecutive tiles is assigned statically; a wave whose tile lies past the end runs masked (clamped loads, no stores).
    const int tiles_per_block = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tile_lo = kQueue ? blockIdx.x * tiles_per_block : 0, tile_hi = kQueue ? min(ntiles, tile_lo + tiles_per_block) : ntiles;
    int *queue = reinterpret_cast<int *>(Wo + kHP + 8);
    int tile = blockIdx.x * wpb + wave;
    if constexpr (kQueue) {
        if (tid == 0) *queue = tile_lo + wpb;
        __syncthreads();
        tile = tile_lo + wave;
    }
    // ---- W1 slabs (W1G == 2): slab sl = W1[:, 48 sl .. 48 sl + 47] as [kHP][kSlabLd] (rows 96..111 = output features 96..99 replicated,
    // the 4x4 blocks of M tile 6; k >= F zero).  slab_load only issues the global loads (one slab AHEAD, in front of the MFMAs of the
    // current slab), slab_store commits them to the other buffer behind those MFMAs; one barrier per slab.
    constexpr int kSlabV = kHP * (kSlabK / 4);                       // float4s per slab
    constexpr int kSlabU = (kSlabV + NTHR - 1) / NTHR;
    f32x4 slab_r[W1G == 2 ? kSlabU : 1];
    int slab_buf = 0;
    auto slab_load = [&](int sl) {
        if constexpr (W1G == 2) {
#pragma unroll
            for (int u = 0; u < kSlabU; ++u) {
                const int i4 = tid + u * NTHR, r = i4 / (kSlabK / 4), c = 4 * (i4 - r * (kSlabK / 4));
                const int wr = r < 96 ? r : 96 + ((r - 96) & 3), k = kSlabK * sl + c;
                slab_r[u] = *reinterpret_cast<const f32x4 *>(P + (size_t)(wr < kHP ? wr : 0) * F + (k < F ? k : 0));
            }
        }
    };
    auto slab_store = [&](int sl, int buf) {
        if constexpr (W1G == 2) {
#pragma unroll
            for (int u = 0; u < kSlabU; ++u) {
                const int i4 = tid + u * NTHR, r = i4 / (kSlabK / 4), c = 4 * (i4 - r * (kSlabK / 4));
                const bool real = kSlabK * sl + c < F;
                f32x4 v = slab_r[u];
                if (!real) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i4 < kSlabV) *reinterpret_cast<f32x4 *>(W1s + (size_t)buf * kHP * kSlabLd + (size_t)r * kSlabLd + c) = v;
            }
        }
    };
    if constexpr (W1G == 2) { slab_load(0); slab_store(0, 0); }
    prefetch_first(tile, xpre);
    for (; (kQueue ? tile : tile - wave) < tile_hi; ++ntile_done) {
        int next_tile = tile_hi;
        auto advance = [&]() {                      // pop the next tile and start its first X loads
            if constexpr (kQueue) {
                int nxt = 0;
                if (lane == 0) nxt = atomicAdd(queue, 1);
                next_tile = __builtin_amdgcn_readfirstlane(nxt);
            } else {
                next_tile = tile + gridDim.x * wpb;
            }
            prefetch_first(next_tile, xpre);
        };
        const int row0 = tile * rows_per_tile;
        const bool tile_full = row0 + rows_per_tile <= R;
        int row[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) row[rt] = row0 + 16 * rt + j;
        FWD_STAMP(0);

        // X loads are branch-free: out-of-range rows / columns read a clamped (valid) address and are zeroed by a select
        const float *xrow[RT];
        bool rok[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            rok[rt] = row[rt] < R;
#ifdef PTR_FWD_XL2      // experiment: the same loads from an L2-resident part of X
            xrow[rt] = X + (size_t)((rok[rt] ? row[rt] : R - 1) & 4095) * F;
#else
            xrow[rt] = X + (size_t)(rok[rt] ? row[rt] : R - 1) * F;
#endif
        }
        // load_raw only issues the loads; finish_x (zero padding + input dropout) runs AFTER the MFMAs of the super-step the
        // loads are prefetched under — anything consuming the loaded value earlier would pull the s_waitcnt in front of them.
        // K TAIL (tq > 0): the contraction dimension is NOT padded to a multiple of 16.  The last super-step holds rem = F - 16 * nSf
        // features; with rem = 4 * tq (tq = 1, 2) lane group g takes features k = 16 nSf + tq g + c, c < tq, so tq MFMAs per output tile
        // cover them instead of four half-empty ones: the lanes load the aligned float4 that holds their features (tq = 2: groups 0/1 the
        // first, 2/3 the second; tq = 1: all the same one) and finish_x picks them out.
        auto load_raw = [&](int S, f32x4 (&xb)[RT]) {
            int k0 = 16 * S + 4 * g;
            if (tq > 0 && S == nSf) k0 = 16 * S + (tq == 2 ? 4 * (g >> 1) : 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (VEC) {
                    xb[rt] = *reinterpret_cast<const f32x4 *>(xrow[rt] + (k0 < F ? k0 : 0));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xb[rt][c] = xrow[rt][k0 + c < F ? k0 + c : 0];
                }
            }
        };
        auto finish_x = [&](int S, f32x4 (&xb)[RT]) {
            const bool tail = tq > 0 && S == nSf;                  // uniform
            int k0 = 16 * S + 4 * g;
            if (tail) k0 = 16 * S + (tq == 2 ? 4 * (g >> 1) : 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 v = xb[rt];
                if ((!tail && 16 * S + 16 > F) || !tile_full) {   // uniform: only a padded last super-step and the tail tile need the zero padding
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] *= ((k0 + c < F) && rok[rt]) ? 1.0f : 0.0f;
                }
                if constexpr (TRAIN) {
                    uint32_t w0, w1;
                    drop_bits(a.seed_lo, a.seed_hi, 0, row[rt], k0 >> 2, w0, w1);
                    v = drop4(v, w0, w1, thr, scale);
                }
                if (tail) {                                        // this lane group's tq features to the front
                    if (tq == 2) { const bool hi = g & 1; v[0] = hi ? v[2] : v[0]; v[1] = hi ? v[3] : v[1]; }
                    else v[0] = g == 0 ? v[0] : (g == 1 ? v[1] : (g == 2 ? v[2] : v[3]));
                }
                xb[rt] = v;
            }
        };

        // Stored row image of tile 6 (columns 96..111 of the [112]-float activation row, one float4 per lane group as for the other
        // tiles): lane group 0 gathers features 96..99 from the four groups (each holds 96 + g in element 0), group 1 writes {ones, 0, 0, 0}
        // — column 100 = 1 is the ones column the fused backward reads db_l from (scorer_bwd.hip) — groups 2, 3 zeros.
        auto tail_store = [&](float v, float ones) -> f32x4 {
            const float h1 = __shfl(v, j + 16, 64), h2 = __shfl(v, j + 32, 64), h3 = __shfl(v, j + 48, 64);
            f32x4 o = f32x4{g == 1 ? ones : 0.0f, 0.0f, 0.0f, 0.0f};
            if (g == 0) o = f32x4{v, h1, h2, h3};
            return o;
        };
        // ---- hidden layer 1: K = F, B operand streamed from HBM (X), software-prefetched one super-step ahead
        // M TILE 6 holds the output features 96..99 only: it is multiplied as 16 blocks of 4 features x 4 documents x 1 k
        // (v_mfma_f32_4x4x1_16b_f32, mma below) — block 4 g + j / 4 = documents 4 (j / 4) .. + 3 at the k index of lane group g — so lane
        // (j, g) accumulates features 96..99 of document j over the k = g (mod 4) classes; tile6_finish() adds the four lane groups and
        // leaves feature 96 + g in accumulator element 0: exactly the B operand of ONE k-step of the next layer (k = 96 + g), i.e. the hidden
        // layers contract over 100 features in 25 k-steps, not 28 (the "100 -> 112" padding).
        auto mma = [&](auto mt_, float aop, float bop, f32x4 c) -> f32x4 {
            if constexpr (decltype(mt_)::value == kMT - 1) return __builtin_amdgcn_mfma_f32_4x4x1f32(aop, bop, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bop, c, 0, 0, 0);
        };
        f32x4 acc[kMT][RT];
        auto tile6_finish = [&]() {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 t = acc[kMT - 1][rt];
#pragma unroll
                for (int c = 0; c < 4; ++c) { t[c] += __shfl_xor(t[c], 16, 64); t[c] += __shfl_xor(t[c], 32, 64); }
                acc[kMT - 1][rt][0] = g == 0 ? t[0] : (g == 1 ? t[1] : (g == 2 ? t[2] : t[3]));
            }
        };
        // One super-step of a layer: acc[mt] += A fragment wa[mt] (k-steps c < nc) x B fragment b.  Tiles go in groups of two / three with the
        // k-step outermost inside a group: consecutive MFMAs write DIFFERENT accumulators (a dependent v_mfma_f32_16x16x4_f32 issues after 40
        // cycles, an independent one after 32; the 4x4 form of tile 6 needs an s_nop between dependent k-steps) while a group's fragments
        // die with it (k-step outermost over all seven tiles keeps 28 fragment registers live and spills in the 16-wave form).
        auto mma_tiles = [&](const f32x4 (&wa)[kMT], const f32x4 (&b)[RT], int nc) {
            auto one = [&]<int MT>(std::integral_constant<int, MT> mt_, int c) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[MT][rt] = mma(mt_, wa[MT][c], b[rt][c], acc[MT][rt]);
            };
            using std::integral_constant;
#ifdef PTR_FWD_MT_OUTER     // experiment builds: the round-2 order (tile outermost, its four k-steps back to back)
            static_for<kMT>([&](auto mt_) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nc) one(mt_, c);
            });
            return;
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < nc) { one(integral_constant<int, 0>{}, c); one(integral_constant<int, 1>{}, c); }
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < nc) { one(integral_constant<int, 2>{}, c); one(integral_constant<int, 3>{}, c); }
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < nc) { one(integral_constant<int, 4>{}, c); one(integral_constant<int, 5>{}, c); one(integral_constant<int, 6>{}, c); }
        };
        auto bias_init = [&](const float *b) {
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) {
                // tile 6: elements = features 96..99, the bias enters once (lane group 0)
                f32x4 b4 = *reinterpret_cast<const f32x4 *>(b + (mt < kMT - 1 ? 16 * mt + 4 * g : 96));
                if (mt == kMT - 1) { const float on = g == 0 ? 1.0f : 0.0f; b4 = b4 * on; }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
            }
        };
        bias_init(Bs);
        // X is prefetched TWO super-steps ahead through three register buffers that rotate by name (the loop is unrolled by three:
        // a v_mov rotation would read the newest, still in-flight loads and put their full latency back on the critical path)
        f32x4 xa[RT], xb[RT], xc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xa[rt] = xpre[rt];
        if (NL == 1) advance();
        finish_x(0, xa);
        // A operands (weight fragments): the 7 fragments of a super-step are read as one batch (see the hidden layers below)
        auto read_w1 = [&](int S, int sin, int mt) -> f32x4 {
            const int k0 = 16 * S + 4 * g;
            if constexpr (W1G == 2) {          // the current slab: super-step sin (0..2) of it
                return *reinterpret_cast<const f32x4 *>(W1s + (size_t)slab_buf * kHP * kSlabLd + (size_t)(16 * mt + j) * kSlabLd + 16 * sin + 4 * g);
            } else if constexpr (W1G == 1) {   // [100][F] in global memory: tile row j of tile mt = feature 16 mt + j, tile 6: feature 96 + (j & 3)
                const int wr = mt < kMT - 1 ? 16 * mt + j : 96 + (j & 3);
                return *reinterpret_cast<const f32x4 *>(P + (size_t)wr * F + (k0 < F ? k0 : 0));
            } else {
                return *reinterpret_cast<const f32x4 *>(W1s + (size_t)(16 * mt + j) * ld1 + k0);
            }
        };
        auto l1_step = [&](int S, int sin, f32x4 (&cur)[RT], f32x4 (&nxt)[RT], f32x4 (&nn)[RT]) {
            load_raw(S + 2 < nS1 ? S + 2 : 0, nn);            // past the end: a valid address, never consumed
            __builtin_amdgcn_sched_barrier(0);                // the loads stay HERE (the scheduler sinks them towards their use)
            f32x4 wa[kMT];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) wa[mt] = read_w1(S, sin, mt);
            if constexpr (RT > 1) __builtin_amdgcn_sched_barrier(0);
            mma_tiles(wa, cur, 4);
            finish_x(S + 1, nxt);                             // S + 1 == nS1: finishes values nobody reads
        };
        // slab mode: one slab = the three super-steps of a rotation group.  Entering: barrier (the slab is complete in LDS and every
        // wave has left the other buffer), then the loads of the NEXT slab (slab 0 of the next tile after the last one); leaving: those
        // loads are committed to the other buffer.
        auto slab_enter = [&](int S) {
            if constexpr (W1G == 2) {
                __syncthreads();
                slab_load(S + 3 < nS1 ? S / 3 + 1 : 0);
            }
        };
        auto slab_leave = [&](int S) {
            if constexpr (W1G == 2) {
                slab_store(S + 3 < nS1 ? S / 3 + 1 : 0, slab_buf ^ 1);
                slab_buf ^= 1;
            }
        };
        // the K tail: tq MFMAs per output tile; A operand = W1[row][16 nSf + tq g + c] straight from the [out][in] layout (W1 in LDS only)
        auto l1_tail = [&](f32x4 (&cur)[RT]) {
            if constexpr (W1G == 0) {
                float wt[kMT][2];
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) {
                    const float *wp = W1s + (size_t)(16 * mt + j) * ld1 + 16 * nSf + tq * g;
                    wt[mt][0] = wp[0];
                    wt[mt][1] = wp[tq - 1];                    // tq == 1: the same element again (unused)
                }
                static_for<kMT>([&](auto mt_) {
                    constexpr int mt = mt_;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        acc[mt][rt] = mma(mt_, wt[mt][0], cur[rt][0], acc[mt][rt]);
                        if (tq == 2) acc[mt][rt] = mma(mt_, wt[mt][1], cur[rt][1], acc[mt][rt]);
                    }
                });
            }
        };
        load_raw(nS1 > 1 ? 1 : 0, xb);
        // super-steps 0 .. nSm-1 by the rotating l1_step; with a K tail the last super-step (nSf) is the tail step
        const int nSm = tq > 0 ? nSf : nS1;
        int S1 = 0;
        for (; S1 + 3 <= nSm; S1 += 3) {                      // branch-free body: the compiler counts the loads in flight exactly
            slab_enter(S1);
            l1_step(S1, 0, xa, xb, xc);
            l1_step(S1 + 1, 1, xb, xc, xa);
            l1_step(S1 + 2, 2, xc, xa, xb);
            slab_leave(S1);
        }
        const int left = nSm - S1;                            // 0, 1 or 2 leftover super-steps; then the tail reads the next buffer in turn
        if (left == 0) {
            if (tq > 0) l1_tail(xa);
        } else if (left == 1) {
            slab_enter(S1);
            l1_step(S1, 0, xa, xb, xc);
            slab_leave(S1);
            if (tq > 0) l1_tail(xb);
        } else {
            slab_enter(S1);
            l1_step(S1, 0, xa, xb, xc);
            l1_step(S1 + 1, 1, xb, xc, xa);
            slab_leave(S1);
            if (tq > 0) l1_tail(xc);
        }
        tile6_finish();

        FWD_STAMP(1);
        // ---- hidden layers 2..NL: B operand = the previous layer's output registers
        for (int l = 1; l < NL; ++l) {
            f32x4 hin[kMT][RT];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 h = acc[mt][rt];
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
                    if constexpr (TRAIN) {
#ifndef PTR_FWD_NOHASH
                        if (mt < kMT - 1) {
                            uint32_t w0, w1;
                            drop_bits(a.seed_lo, a.seed_hi, l, row[rt], 4 * mt + g, w0, w1);
                            h = drop4(h, w0, w1, thr, scale);
                        } else {                                 // tile 6: element 0 is feature 96 + g
                            h[0] *= drop_keep1(a.seed_lo, a.seed_hi, l, row[rt], 96 + g, thr) ? scale : 0.0f;
                        }
#endif
                    }
                    hin[mt][rt] = h;
                }
            if (l == NL - 1) advance();
            const float *Wl = Wh + (size_t)(l - 1) * kHP * kH;
            bias_init(Bs + (size_t)l * kHP);
            // the 7 weight fragments of a super-step are read as ONE batch (one exposed LDS latency per 56 MFMAs): with the read issued
            // right in front of its 8 MFMAs — what hipcc schedules on its own — every group waits out the full LDS latency
#pragma unroll
            for (int S = 0; S < kMT; ++S) {
                constexpr int kLast = kMT - 1;
                f32x4 wa[kMT];
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) {
                    const float *wp = Wl + (size_t)(16 * mt + j) * kH;
                    if (S < kLast) wa[mt] = *reinterpret_cast<const f32x4 *>(wp + 16 * S + 4 * g);
                    else wa[mt] = f32x4{wp[96 + g], 0.0f, 0.0f, 0.0f};                      // the K tail: one k-step, k = 96 + g
                }
                if constexpr (RT > 1) __builtin_amdgcn_sched_barrier(0);   // keep the batch in front of the MFMAs (the scheduler would sink every
                                                                           // read to its use); 16-row tiles: 128 VGPRs cannot hold a batch
                if constexpr (TRAIN) {
                    // the layer input is stored for the backward pass one feature tile per super-step, spread over the layer's MFMAs
                    // (28 stores issued back to back at the layer transition stall the wave on the store path)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (row[rt] < R16) {            // whole row tiles (tile-major `acts`, ptr_mlp.h): rows past R hold finite values nobody uses
#if defined(PTR_FWD_L2STORE)     // experiment: same store instructions, L2-resident target
                            float *arow = acts + (size_t)(l - 1) * act_layer_floats(R) + act_off(row[rt] & 4095, 0);
#else
                            float *arow = acts + (size_t)(l - 1) * act_layer_floats(R) + act_off(row[rt], 0);
#endif
#if !defined(PTR_FWD_NOSTORE)
                            *reinterpret_cast<f32x4 *>(arow + 256 * S + 4 * g) = S < kLast ? hin[S][rt] : tail_store(hin[S][rt][0], 1.0f);
#else
                            if (hin[S][rt][0] == 123.456f) acts[0] = 1.0f;
#endif
                        }
                    }
                    if constexpr (RT > 1) __builtin_amdgcn_sched_barrier(0);
                }
                mma_tiles(wa, hin[S], S < kLast ? 4 : 1);
            }
            tile6_finish();
        }

        FWD_STAMP(2);
        // ---- last hidden activation + output layer (100 -> 1): VALU dot product, reduced over the 4 lane groups
        float sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) sc[rt] = 0.0f;
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wo + 16 * mt + 4 * g);      // tile 6: {w_out[96 + g], 0, 0, 0}
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 h = acc[mt][rt];
#pragma unroll
                for (int c = 0; c < (mt == kMT - 1 ? 1 : 4); ++c) { h[c] = fmaxf(h[c], 0.0f); sc[rt] = fmaf(h[c], w4[c], sc[rt]); }
                if constexpr (TRAIN) {
                    if (row[rt] < R16) {
#if defined(PTR_FWD_L2STORE)
                        float *arow = acts + (size_t)(NL - 1) * act_layer_floats(R) + act_off(row[rt] & 4095, 0);
#else
                        float *arow = acts + (size_t)(NL - 1) * act_layer_floats(R) + act_off(row[rt], 0);
#endif
#if defined(PTR_FWD_NOSTORE)
                        if (h[0] == 123.456f) acts[1] = 1.0f;
#else
                        *reinterpret_cast<f32x4 *>(arow + 256 * mt + 4 * g) = mt < kMT - 1 ? h : tail_store(h[0], 0.0f);
#endif
                    }
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float s = sc[rt];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (g == 0 && row[rt] < R) preds[row[rt]] = s + b_out;
        }
        FWD_STAMP(3);
        tile = next_tile;
    }
}

// =================================================================================================== backward: dZ chain
// LDS: WT (NL-1) x [kHP][kH] (hidden weights TRANSPOSED: WT[k][out]) | Wo [kHP] | red [kHP + 16]
__host__ __device__ inline size_t dz_lds_floats(int NL) { return (size_t)(NL - 1) * kHP * kH + (kHP + 16) + 16 * (kHP + 16); }

// dz[l] (l = 0..NL-1) = dLoss/d(pre-activation of hidden layer l+1), [NL][R][100].  acts[l] = post-dropout input of hidden
// layer l+2 (l < NL-1) / last hidden activation (l = NL-1), as the forward kernel stored them.
// ws[block][wout_off ..]: per-block partial of d w_out (100) and d b_out, in the flat parameter layout.
template <int RT, int NTHR>
__global__ void __launch_bounds__(NTHR)
mlp_bwd_dz_kernel(const float *__restrict__ P, const float *__restrict__ acts, const float *__restrict__ dpreds, MlpArgs a,
                  float *__restrict__ dz, float *__restrict__ ws, size_t np_stride, size_t wout_off) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = a.F, NL = a.NL, R = a.R;
    float *WT = smem;
    float *Wo = WT + (size_t)(NL - 1) * kHP * kH;
    float *wrow = Wo + kHP + 16;                          // [<= 16 waves][kHP + 16] partial d w_out / d b_out
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int l = 1; l < NL; ++l) stage_matrix(WT + (size_t)(l - 1) * kHP * kH, kH, P + off_W(l, F), kH, kH, true, tid, nthr);
    stage_vector(Wo, P + off_wout(NL, F), kH, tid, nthr);
    if (tid < 16) Wo[kHP + tid] = 0.0f;
    __syncthreads();

    const int lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wpb = nthr >> 6, wave = tid >> 6;
    const int rows_per_tile = 16 * RT;
    const int ntiles = (R + rows_per_tile - 1) / rows_per_tile;
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    f32x4 dwo[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) dwo[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.0f;

    // The tile's first operands (last hidden activation, dpreds) are prefetched one tile ahead as RAW loads (clamped
    // addresses, no consumer until the next iteration), so their HBM latency hides behind the previous tile's MFMA chain.
    f32x4 hpf[kMT][RT];
    float dspf[RT];
    auto prefetch_top = [&](int tile) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int r = tile * rows_per_tile + 16 * rt + j;
            const int rc = r < R ? r : R - 1;
            dspf[rt] = dpreds[rc];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
                hpf[mt][rt] = *reinterpret_cast<const f32x4 *>(acts + (size_t)(NL - 1) * act_layer_floats(R) + act_off(rc, 16 * mt + 4 * g));
        }
    };
    const int tile_first = blockIdx.x * wpb + wave, tile_step = gridDim.x * wpb;
    if (tile_first < ntiles) prefetch_top(tile_first);
    for (int tile = tile_first; tile < ntiles; tile += tile_step) {
        const int row0 = tile * rows_per_tile;
        int row[RT];
        float ds[RT];
        f32x4 htop[kMT][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            row[rt] = row0 + 16 * rt + j;
            ds[rt] = dspf[rt] * (row[rt] < R ? 1.0f : 0.0f);
            if (g == 0) dbo += ds[rt];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) htop[mt][rt] = hpf[mt][rt];
        }
        if (tile + tile_step < ntiles) prefetch_top(tile + tile_step);
        // top: dz_{NL-1} = ds * w_out * [h > 0];  d w_out += h * ds
        f32x4 cur[kMT][RT];
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wo + 16 * mt + 4 * g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bool ok = row[rt] < R;
                const size_t o = ((size_t)(NL - 1) * R + row[rt]) * kAL + 16 * mt + 4 * g;
                f32x4 h = htop[mt][rt];
                const float okf = ok ? 1.0f : 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] *= okf;
                f32x4 d;
#pragma unroll
                for (int c = 0; c < 4; ++c) { d[c] = (ds[rt] * w4[c]) * (h[c] > 0.0f ? 1.0f : 0.0f); dwo[mt][c] = fmaf(h[c], ds[rt], dwo[mt][c]); }
                if (ok) *reinterpret_cast<f32x4 *>(dz + o) = d;
                cur[mt][rt] = d;
            }
        }
        for (int l = NL - 1; l >= 1; --l) {
            // dA_{l-1}^T[k][row] = sum_out W_l[out][k] * dz_l[row][out]
            const float *Wl = WT + (size_t)(l - 1) * kHP * kH;
            // the gate (stored post-dropout activation of layer l-1) is fetched BEFORE the MFMA chain so that its HBM latency
            // hides under the 7x7 tile products
            f32x4 gate[kMT][RT];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bool ok = row[rt] < R;
                    const size_t o = (size_t)(l - 1) * act_layer_floats(R) + act_off(row[rt], 16 * mt + 4 * g);       // (`acts` is tile-major; dz row-major)
                    gate[mt][rt] = *reinterpret_cast<const f32x4 *>(acts + (ok ? o : 0));
                }
            f32x4 acc[kMT][RT];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // per super-step: the 7 weight fragments as one batch of LDS reads, then the tiles in groups of two / three with the k-step
            // outermost inside a group — consecutive MFMAs write different accumulators (dependent 16x16x4: 40 cycles, independent: 32)
#pragma unroll
            for (int S = 0; S < kMT; ++S) {
                f32x4 wa[kMT];
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) wa[mt] = *reinterpret_cast<const f32x4 *>(Wl + (size_t)(16 * mt + j) * kH + 16 * S + 4 * g);
                auto group = [&](int m0, int m1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int mt = m0; mt < m1; ++mt)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[mt][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][c], cur[S][rt][c], acc[mt][rt], 0, 0, 0);
                };
                group(0, 2); group(2, 4); group(4, kMT);
            }
            // gate: a > 0  <=>  kept by dropout AND relu active
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bool ok = row[rt] < R;
                    const size_t o = ((size_t)(l - 1) * R + row[rt]) * kAL + 16 * mt + 4 * g;
                    f32x4 d;
#pragma unroll
                    for (int c = 0; c < 4; ++c) d[c] = acc[mt][rt][c] * ((ok && gate[mt][rt][c] > 0.0f) ? inv_keep : 0.0f);
                    if (ok) *reinterpret_cast<f32x4 *>(dz + o) = d;
                    cur[mt][rt] = d;
                }
        }
    }
    // block partial of d w_out / d b_out: butterfly over the 16 row-lanes j, one LDS row per wave, then a fixed-order sum
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = dwo[mt][c];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (j == 0) wrow[(size_t)wave * (kHP + 16) + 16 * mt + 4 * g + c] = v;
        }
    {
        float v = dbo;
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (lane == 0) wrow[(size_t)wave * (kHP + 16) + kHP] = v;
    }
    __syncthreads();
    for (int i = tid; i < kH + 1; i += nthr) {
        const int src = i < kH ? i : kHP;                      // d b_out sits behind the padded d w_out row
        float s = 0.0f;
        for (int w = 0; w < wpb; ++w) s += wrow[(size_t)w * (kHP + 16) + src];
        ws[(size_t)blockIdx.x * np_stride + wout_off + i] = s;
    }
}

// =================================================================================================== backward: dW (row contraction)
// One launch per layer.  dW[out][in] = sum_rows dZ[row][out] * A[row][in];  db[out] = sum_rows dZ[row][out].
// A = X with the input dropout recomputed (layer 0, SITE0) or the stored activation (other layers).  Each wave owns the
// in-feature tiles nt = wave, wave + 4, ... and all 7 out-feature tiles; a block walks its contiguous chunk of rows 4 at a time.
// ws[block][n_params]: per-block partial gradient in the flat parameter layout, reduced by reduce_partials_kernel.
template <int NTW, bool SITE0, int NWV>
__global__ void __launch_bounds__(NWV * 64)
mlp_bwd_dw_kernel(const float *__restrict__ A, int lda, const float *__restrict__ dZ, int K, int nt_base, MlpArgs a,
                  float *__restrict__ ws, size_t np_stride, size_t w_off, size_t b_off) {
    const int R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int chunk = ((R + gridDim.x - 1) / gridDim.x + 4 * DW_U - 1) / (4 * DW_U) * (4 * DW_U);
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const uint32_t thr = drop_thr(a.p_drop);
    const float scale = (SITE0 && a.p_drop > 0.0f) ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int ntk = (K + 15) >> 4;
    (void)ntk;

    f32x4 acc[NTW][kMT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbv[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) dbv[mt] = 0.0f;

    constexpr int U = DW_U;                                // k-steps (of 4 rows) per iteration (prefetch depth)
    // Per-lane column offsets / validity are loop invariant; loads are branch-free (clamped column, select-to-zero).
    int fa[kMT], kb[NTW];
    bool fa_ok[kMT], kb_ok[NTW];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) { const int f = 16 * mt + j; fa_ok[mt] = true; fa[mt] = f; }   // padded: always in range
#pragma unroll
    for (int t = 0; t < NTW; ++t) { const int k = 16 * (nt_base + wave + NWV * t) + j; kb_ok[t] = k < K; kb[t] = kb_ok[t] ? k : 0; }
    float av[U][kMT], bv[U][NTW], avn[U][kMT], bvn[U][NTW];
    auto load = [&](int r0, bool guard, float (&pa)[U][kMT], float (&pb)[U][NTW]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 4 * u + g;
            const bool rok = !guard || r < r_end;
            const int rc = rok ? r : r_end - 1;
            const float *dzr = dZ + (size_t)rc * kAL;
            // A = X (row-major, lda) for the first layer, the stored activations (tile-major, ptr_mlp.h act_off) for the others
            const float *ar = SITE0 ? A + (size_t)rc * lda : A + act_off(rc, 0);
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) pa[u][mt] = dzr[fa[mt]] * ((rok & fa_ok[mt]) ? 1.0f : 0.0f);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                float v = SITE0 ? ar[kb[t]] : ar[((kb[t] >> 4) << 8) + (kb[t] & 15)];
                bool keep = rok & kb_ok[t];
                if constexpr (SITE0) keep = keep & drop_keep1(a.seed_lo, a.seed_hi, 0, rc, kb[t], thr);   // thr == 0 keeps all
                pb[u][t] = v * (keep ? scale : 0.0f);
            }
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) dbv[mt] += av[u][mt];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][t], acc[t][mt], 0, 0, 0);
        }
    };
    const int n_full = r_end > r_begin ? (r_end - r_begin) / (4 * U) : 0;     // iterations with every row valid
    int r0 = r_begin;
    if (n_full > 0) {
        load(r0, false, av, bv);
        for (int it = 0; it < n_full; ++it, r0 += 4 * U) {
            if (it + 1 < n_full) load(r0 + 4 * U, false, avn, bvn);
            mma();
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt) av[u][mt] = avn[u][mt];
#pragma unroll
                for (int t = 0; t < NTW; ++t) bv[u][t] = bvn[u][t];
            }
        }
    }
    if (r0 < r_end) { load(r0, true, av, bv); mma(); }               // guarded tail (< 16 rows)
    float *out = ws + (size_t)blockIdx.x * np_stride + w_off;
    float *outb = ws + (size_t)blockIdx.x * np_stride + b_off;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nt = nt_base + wave + NWV * t, k = 16 * nt + j;
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int o = 16 * mt + 4 * g + c;
                if (nt < ntk && k < K && o < kH) out[(size_t)o * K + k] = acc[t][mt][c];
            }
    }
    if (wave == 0 && nt_base == 0) {
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            float v = dbv[mt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int f = 16 * mt + j;
            if (g == 0 && f < kH) outb[f] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-staged variant of the dW kernel (used whenever rows are 16-byte aligned, i.e. lda % 4 == 0).
// The direct variant above lets all four waves of a block fetch the same dZ tiles (4x redundant, dword-wide loads): only
// ~1/4 of the bytes in flight are unique and the kernel sits at 1.8-2.8 TB/s.  Here the workgroup loads each 16-row
// slab of dZ and of its A column slice ONCE with 16-byte loads (one dropout hash per float4 for the input site), double
// buffers it in LDS, and the waves read their MFMA operands (lane j = feature, g = row) back with ds_read_b32; row strides
// are = 16 (mod 32) floats so the two 16-lane halves of a read hit disjoint banks.
// One pass covers the A columns [64*NTW*pass, 64*NTW*(pass+1)); wave w owns in-feature tiles w, w+4, ...
template <int NTW, bool SITE0, int RB>
__global__ void __launch_bounds__(256)
mlp_bwd_dw_lds_kernel(const float *__restrict__ A, int lda, const float *__restrict__ dZ, int K, int nt_base, MlpArgs a,
                      float *__restrict__ ws, size_t np_stride, size_t w_off, size_t b_off) {
    // RB = rows per slab (RB/4 MFMA k-steps per barrier)
    constexpr int WA4 = 16 * NTW;                  // float4 per A-slice row (64*NTW columns)
    constexpr int LDA = 64 * NTW + 16;             // LDS row strides: = 16 (mod 32)
    constexpr int LDZ = kAL;                       // 112 = 16 (mod 32)
    constexpr int SA = RB * WA4 / 256;             // float4 load slots per thread for the A slab (NTW)
    constexpr int SZ = (RB * (kAL / 4) + 255) / 256;   // ... for the dZ slab (2, second one half used)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto zb = [&](int b) -> float * { return smem + b * (RB * LDZ); };
    auto ab = [&](int b) -> float * { return smem + 2 * RB * LDZ + b * (RB * LDA); };

    const int R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int chunk = ((R + gridDim.x - 1) / gridDim.x + RB - 1) / RB * RB;
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const uint32_t thr = drop_thr(a.p_drop);
    const float scale = (SITE0 && a.p_drop > 0.0f) ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int col0 = 16 * nt_base;                 // first A column of this pass

    // loop-invariant slot geometry
    int a_row[SA], a_col[SA], z_row[SZ], z_col[SZ];
    bool a_ok[SA], z_ok[SZ];
#pragma unroll
    for (int s_ = 0; s_ < SA; ++s_) {
        const int idx = s_ * 256 + tid;
        a_row[s_] = idx / WA4;
        a_col[s_] = col0 + 4 * (idx % WA4);
        a_ok[s_] = a_col[s_] < K;
    }
#pragma unroll
    for (int s_ = 0; s_ < SZ; ++s_) {
        const int idx = s_ * 256 + tid;
        z_ok[s_] = idx < RB * (kAL / 4);
        z_row[s_] = z_ok[s_] ? idx / (kAL / 4) : 0;
        z_col[s_] = z_ok[s_] ? 4 * (idx % (kAL / 4)) : 0;
    }
    f32x4 ra[SA], rz[SZ];
    // global -> registers: RAW loads from clamped (always valid) addresses.  Nothing here consumes the values — the zero
    // padding and the recomputed input dropout are applied in lstore(), i.e. AFTER the MFMAs of the slab these loads are
    // prefetched under; a consumer here would pull the s_waitcnt in front of them.
    auto gload = [&](int r0) {
#pragma unroll
        for (int s_ = 0; s_ < SA; ++s_) {
            const int r = r0 + a_row[s_];
            const int rc = r < r_end ? r : r_end - 1;
            const int cc = a_ok[s_] ? a_col[s_] : 0;
            ra[s_] = *reinterpret_cast<const f32x4 *>(SITE0 ? A + (size_t)rc * lda + cc : A + act_off(rc, cc));      // X row-major / activations tile-major
        }
#pragma unroll
        for (int s_ = 0; s_ < SZ; ++s_) {
            const int r = r0 + z_row[s_];
            const int rc = r < r_end ? r : r_end - 1;
            rz[s_] = *reinterpret_cast<const f32x4 *>(dZ + (size_t)rc * kAL + z_col[s_]);
        }
    };
    auto lstore = [&](int buf, int r0) {            // registers -> LDS slab (rows r0 ..), masks applied here
#pragma unroll
        for (int s_ = 0; s_ < SA; ++s_) {
            const int r = r0 + a_row[s_];
            const bool rok = r < r_end;
            f32x4 v = ra[s_];
            if constexpr (SITE0) {
                uint32_t w0, w1;
                drop_bits(a.seed_lo, a.seed_hi, 0, rok ? r : r_end - 1, a_col[s_] >> 2, w0, w1);
                v = drop4(v, w0, w1, thr, scale);
            }
            const float okf = (rok & a_ok[s_]) ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= okf;
            *reinterpret_cast<f32x4 *>(ab(buf) + a_row[s_] * LDA + (a_col[s_] - col0)) = v;
        }
#pragma unroll
        for (int s_ = 0; s_ < SZ; ++s_) {
            if (z_ok[s_]) {
                const float okf = (r0 + z_row[s_] < r_end) ? 1.0f : 0.0f;
                f32x4 v = rz[s_];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] *= okf;
                *reinterpret_cast<f32x4 *>(zb(buf) + z_row[s_] * LDZ + z_col[s_]) = v;
            }
        }
    };

    f32x4 acc[NTW][kMT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbv[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) dbv[mt] = 0.0f;

    if (r_begin < r_end) {
        gload(r_begin);
        lstore(0, r_begin);
    }
    __syncthreads();
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += RB, buf ^= 1) {
        const bool more = r0 + RB < r_end;
        if (more) gload(r0 + RB);                   // next slab in flight while this one is multiplied
        const float *zs = zb(buf), *as = ab(buf);
#pragma unroll
        for (int u = 0; u < RB / 4; ++u) {
            float av[kMT], bv[NTW];
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) { av[mt] = zs[(4 * u + g) * LDZ + 16 * mt + j]; dbv[mt] += av[mt]; }
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[t] = as[(4 * u + g) * LDA + 16 * (wave + 4 * t) + j];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int mt = 0; mt < kMT; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], acc[t][mt], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1, r0 + RB);
        __syncthreads();
    }
    float *out = ws + (size_t)blockIdx.x * np_stride + w_off;
    float *outb = ws + (size_t)blockIdx.x * np_stride + b_off;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int k = col0 + 16 * (wave + 4 * t) + j;
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int o = 16 * mt + 4 * g + c;
                if (k < K && o < kH) out[(size_t)o * K + k] = acc[t][mt][c];
            }
    }
    if (wave == 0 && nt_base == 0) {
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            float v = dbv[mt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int f = 16 * mt + j;
            if (g == 0 && f < kH) outb[f] = v;
        }
    }
}
__host__ __device__ constexpr size_t dw_lds_bytes(int NTW, int RB) { return (size_t)(2 * RB * kAL + 2 * RB * (64 * NTW + 16)) * sizeof(float); }

// grad[i] = sum_b ws[b][i], fixed order.  A 1024-thread workgroup owns 64 consecutive i: wave w sums the partials
// b = w, w+16, ... (4 independent accumulators keep 4 loads in flight), then one wave adds the 16 wave totals in order.
// Entries i >= tail_begin (d w_out, d b_out: written by the dZ kernel's smaller grid) only have nblk_tail partials.
// Optional epilogue of the partial reduction: the optimiser step on the element just reduced (single-device training: no all-reduce sits
// between gradient and step) and the sum of the loss slots of the batch — three launches fewer per train step.
struct OptStep {
    int kind;                      // 0 = none, PTR_OPT_ADAM / _ADAGRAD / _RMSPROP
    float lr, h1, h2, eps, wd, bc1, bc2_sqrt;
    float *param, *s1, *s2;
    const float *loss_q; int nq; float *loss_out;
    uint8_t *wimg; int F, NL;      // r6: the bf16x6 forward's weight image, refreshed element by element behind the step (NULL: none)
};
__global__ void __launch_bounds__(1024)
reduce_partials_kernel(const float *__restrict__ ws, int nblk, int nblk_tail, size_t tail_begin, size_t stride, size_t n,
                       float *__restrict__ grad, OptStep o) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (o.loss_out && blockIdx.x == gridDim.x - 1) {
        // one extra block: the loss slots of the batch, summed by the function sum_f32_kernel calls (ptr_device.h block1024_sum): the fused
        // step returns the same bits as the separate call
        const float tot = block1024_sum(o.loss_q, o.nq, &part[0][0]);
        if (threadIdx.x == 0) o.loss_out[0] = tot;
        return;
    }
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool ok = i < n;
    const size_t ic = ok ? i : n - 1;
    const int nb = ic >= tail_begin ? nblk_tail : nblk;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = w;
    for (; b + 48 < nb; b += 64) {
        s0 += ws[(size_t)b * stride + ic];
        s1 += ws[(size_t)(b + 16) * stride + ic];
        s2 += ws[(size_t)(b + 32) * stride + ic];
        s3 += ws[(size_t)(b + 48) * stride + ic];
    }
    for (; b < nb; b += 16) s0 += ws[(size_t)b * stride + ic];
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && ok) {
        float s = part[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) s += part[k][lane];
        if (grad) grad[i] = s;                          // (ptr_opt_step_loss: ws IS the gradient, one "partial" — nothing to write back)
        // fused optimiser step on the element just reduced: the arithmetic of adam_kernel / adagrad_kernel / rmsprop_kernel, bit for bit
        if (o.kind == PTR_OPT_ADAM) {
            const float pi = o.param[i];
            const float gi = s + o.wd * pi;
            const float mi = o.h1 * o.s1[i] + (1.0f - o.h1) * gi;
            const float vi = o.h2 * o.s2[i] + (1.0f - o.h2) * gi * gi;
            o.s1[i] = mi; o.s2[i] = vi;
            const float denom = sqrtf(vi) / o.bc2_sqrt + o.eps;
            const float pn = pi - (o.lr / o.bc1) * (mi / denom);
            o.param[i] = pn;
            if (o.wimg) x6_img_put(o.wimg, o.F, o.NL, i, pn);
        } else if (o.kind == PTR_OPT_ADAGRAD) {
            const float pi = o.param[i];
            const float gi = s + o.wd * pi;
            const float si = o.s1[i] + gi * gi;
            o.s1[i] = si;
            const float pn = pi - o.lr * (gi / (sqrtf(si) + o.eps));     // o.lr = the decayed clr
            o.param[i] = pn;
            if (o.wimg) x6_img_put(o.wimg, o.F, o.NL, i, pn);
        } else if (o.kind == PTR_OPT_RMSPROP) {
            const float pi = o.param[i];
            const float gi = s + o.wd * pi;
            const float si = o.h1 * o.s1[i] + (1.0f - o.h1) * gi * gi;
            o.s1[i] = si;
            const float pn = pi - o.lr * (gi / (sqrtf(si) + o.eps));
            o.param[i] = pn;
            if (o.wimg) x6_img_put(o.wimg, o.F, o.NL, i, pn);
        }
    }
}

// =================================================================================================== Adam (torch.optim.Adam semantics)
// g = grad + wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(256)
adam_kernel(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, size_t n, float lr,
            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float pi = p[i];
    const float gi = grad[i] + wd * pi;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

// debug / test helper: the dropout keep-mask of one site as floats [R][n_feat]
__global__ void __launch_bounds__(256)
dropout_mask_kernel(MlpArgs a, int site, int n_feat, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.R * n_feat) return;
    const int row = (int)(i / n_feat), k = (int)(i % n_feat);
    out[i] = drop_keep1(a.seed_lo, a.seed_hi, site, row, k, drop_thr(a.p_drop)) ? 1.0f : 0.0f;
}

static int check_mlp(const char *who, int R, int F, int NL, float p) {
    if (R < 0 || F <= 0 || NL < 1 || NL > kMaxLayers) { set_error("%s: bad shape R=%d F=%d NL=%d", who, R, F, NL); return PTR_ERR_INVALID_ARG; }
    if (!(p >= 0.0f && p < 1.0f)) { set_error("%s: dropout p=%g out of [0,1)", who, (double)p); return PTR_ERR_INVALID_ARG; }
    const int w1g = fwd_w1_mode(F, NL);
    if (fwd_lds_floats(F, NL, w1g) * sizeof(float) > 160 * 1024 || (w1g && F % 4 != 0) || (F + 15) / 16 > 48) {
        set_error("%s: F=%d with %d hidden layers is outside the fused scorer's range (LDS %zu KB, max 160; F %% 4 == 0 needed "
                  "above ~157 features; F <= 768)", who, F, NL, fwd_lds_floats(F, NL, w1g) * sizeof(float) / 1024);
        return PTR_ERR_UNSUPPORTED;
    }
    return 0;
}

static int dw_blocks_per_cu() {
    static int v = 0;
    if (!v) { const char *e = getenv("PTR_DW_BLOCKS_PER_CU"); v = e ? atoi(e) : 2; if (v < 1 || v > 8) v = 2; }
    return v;
}

// Tile-shape knobs (measured on MI355X, B*L = 524288 rows, F = 136): forward 16 waves x 16-row tiles 432 us vs 8 waves x
// 32-row tiles 461 us; dZ 8 waves 400 us vs 16 waves (spills at the 128-VGPR cap) 415 us.
static int env_flag(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? (atoi(e) != 0) : dflt;
}
static int fwd_wide() { static int v = -2; if (v == -2) v = env_flag("PTR_FWD_WIDE", -1); return v; }   // -1: by configuration
static int dw_staged() { static int v = -1; if (v < 0) v = env_flag("PTR_DW_STAGED", 1); return v; }
static int dw_rb() { static int v = -1; if (v < 0) { const char *e = getenv("PTR_DW_RB"); v = (e && atoi(e) == 32) ? 32 : 16; } return v; }   // 16 measured best (32: 1.09 vs 1.05 ms backward)
static int dz_wide() { static int v = -1; if (v < 0) v = env_flag("PTR_DZ_WIDE", 0); return v; }

int mlp_num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace ptr

extern "C" size_t ptr_mlp_num_params(int F, int NL) { return ptr::n_params(NL, F); }

// floats of workspace ptr_mlp_backward needs
extern "C" size_t ptr_mlp_backward_ws_floats(int F, int NL) {
    return (size_t)ptr::dw_blocks_per_cu() * ptr::mlp_num_cus() * ptr::n_params(NL, F);
}

// floats of dZ scratch ptr_mlp_backward needs for (R, F, NL): 0 when the single-pass fused backward serves the configuration
// (X / acts assumed 16-byte aligned, as every torch allocation is)
extern "C" size_t ptr_mlp_acts_floats(int R, int NL) { return R > 0 && NL > 0 ? (size_t)NL * ptr::act_layer_floats(R) : 0; }
extern "C" size_t ptr_mlp_backward_dz_floats(int R, int F, int NL) {
    return ptr::bwd_fused_supported(F, NL, nullptr, nullptr) ? 0 : (size_t)NL * (size_t)R * ptr::kAL;
}

extern "C" int ptr_mlp_forward(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed,
                               float *preds, float *acts, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mlp_forward";
    if (int rc = check_mlp(who, R, F, NL, p_drop)) return rc;
    if (R > 0 && (!X || !params || !preds || (train && !acts))) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (R == 0) return 0;
    MlpArgs a{R, F, NL, train ? p_drop : 0.0f, (uint32_t)seed, (uint32_t)(seed >> 32)};
    const int w1g = fwd_w1_mode(F, NL);
    const size_t lds = fwd_lds_floats(F, NL, w1g) * sizeof(float);
    const bool vec = (F % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    // 16 waves x 16-row tiles (4 waves/SIMD) or 8 waves x 32-row tiles (half the weight-fragment LDS reads per MFMA).  Measured
    // inside the train step (rocprofv3, F=136): the 16-row form is 7-9 % faster from 1024 to 4096 queries of 128 documents and
    // 20-30 % faster below 256 (twice the waves on few tiles); in eval mode and with first-layer weights streamed from L2
    // (F=700) the 32-row form wins.  PTR_FWD_WIDE=0/1 pins the choice.
    const bool wide = fwd_wide() < 0 ? (train && !w1g) : fwd_wide() != 0;
    const int rows_per_tile = wide ? 16 : 32, wpb = wide ? 16 : 8;
    const int ntiles = (R + rows_per_tile - 1) / rows_per_tile;
    const int grid = ntiles < mlp_num_cus() ? ntiles : mlp_num_cus();      // few tiles: one per CU (the SIMD to itself) before two per CU
    auto launch = [&](auto kern) -> int {
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(grid > 0 ? grid : 1), dim3(wpb * 64), lds, as_stream(stream), X, params, a, preds, acts);
        return check_hip(hipGetLastError(), who);
    };
    if (w1g) {   // large F (e.g. Yahoo's 700): W1 streamed from L2, F % 4 == 0 guaranteed by check_mlp
        if (!vec) { set_error("%s: X must be 16-byte aligned for F=%d", who, F); return PTR_ERR_INVALID_ARG; }
        const int force_stream = env_flag("PTR_FWD_W1_STREAM", 0);             // tests / experiments: the per-wave streaming form (read per call)
        if (w1g == 2 && !force_stream) {
            if (wide) return train ? launch(mlp_fwd_kernel<1, 1024, true, true, 2, 0>) : launch(mlp_fwd_kernel<1, 1024, false, true, 2, 0>);
            return train ? launch(mlp_fwd_kernel<2, 512, true, true, 2, 0>) : launch(mlp_fwd_kernel<2, 512, false, true, 2, 0>);
        }
        if (wide) return train ? launch(mlp_fwd_kernel<1, 1024, true, true, 1, 0>) : launch(mlp_fwd_kernel<1, 1024, false, true, 1, 0>);
        return train ? launch(mlp_fwd_kernel<2, 512, true, true, 1, 0>) : launch(mlp_fwd_kernel<2, 512, false, true, 1, 0>);
    }
    // K tail of layer 1 (F mod 16 of 4 or 8 with float4 X loads): no zero-padded k-steps — 136 features: 34 k-steps instead of 36.
    // Measured (B = 4096 x 128 x 136, r3): the hidden-layer K tails alone 383 -> 370 us; with the layer-1 tail on top 373 us — its selects
    // and tail operands push the 16-wave form past 128 VGPRs (18 spills vs 6).  It is therefore opt-in (PTR_FWD_TQ=1; tests run both).
    int tq = 0;
    if (const char *e = getenv("PTR_FWD_TQ")) { if (atoi(e) != 0 && vec) tq = (F & 15) == 8 ? 2 : ((F & 15) == 4 ? 1 : 0); }
    auto pick = [&](auto rt_, auto nthr_, auto train_) -> int {
        constexpr int RT_ = decltype(rt_)::value, NT_ = decltype(nthr_)::value;
        constexpr bool TR_ = decltype(train_)::value;
        if (!vec) return launch(mlp_fwd_kernel<RT_, NT_, TR_, false, 0, 0>);
        if (tq == 2) return launch(mlp_fwd_kernel<RT_, NT_, TR_, true, 0, 2>);
        if (tq == 1) return launch(mlp_fwd_kernel<RT_, NT_, TR_, true, 0, 1>);
        return launch(mlp_fwd_kernel<RT_, NT_, TR_, true, 0, 0>);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using N1024 = std::integral_constant<int, 1024>; using N512 = std::integral_constant<int, 512>;
    if (wide) return train ? pick(I1{}, N1024{}, std::true_type{}) : pick(I1{}, N1024{}, std::false_type{});
    return train ? pick(I2{}, N512{}, std::true_type{}) : pick(I2{}, N512{}, std::false_type{});
}

namespace ptr {
int mlp_backward_impl(const char *who, const float *X, const float *params, const float *acts, const float *dpreds, int R, int F, int NL,
                             float p_drop, uint64_t seed, float *dz, float *ws, float *grad, void *stream, const OptStep &opt);
}
extern "C" int ptr_mlp_backward(const float *X, const float *params, const float *acts, const float *dpreds, int R, int F, int NL,
                                float p_drop, uint64_t seed, float *dz, float *ws, float *grad, void *stream) {
    return ptr::mlp_backward_impl("ptr_mlp_backward", X, params, acts, dpreds, R, F, NL, p_drop, seed, dz, ws, grad, stream, ptr::OptStep{});
}

extern "C" int ptr_mlp_backward_step(const float *X, float *params, const float *acts, const float *dpreds, int R, int F, int NL, float p_drop,
                                     uint64_t seed, float *dz, float *ws, float *grad, int opt_kind, float lr, float hyper1, float hyper2,
                                     float eps, float weight_decay, int step, float *state1, float *state2, const float *loss_q, int nq,
                                     float *loss_out, void *stream) {
    return ptr::mlp_backward_step_impl(X, params, acts, dpreds, R, F, NL, p_drop, seed, dz, ws, grad, opt_kind, lr, hyper1, hyper2, eps, weight_decay, step, state1,
                                       state2, loss_q, nq, loss_out, nullptr, stream);
}

int ptr::mlp_backward_step_impl(const float *X, float *params, const float *acts, const float *dpreds, int R, int F, int NL, float p_drop, uint64_t seed,
                                float *dz, float *ws, float *grad, int opt_kind, float lr, float hyper1, float hyper2, float eps, float weight_decay, int step,
                                float *state1, float *state2, const float *loss_q, int nq, float *loss_out, void *wimg, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mlp_backward_step";
    if (opt_kind < PTR_OPT_ADAM || opt_kind > PTR_OPT_RMSPROP) { set_error("%s: unknown optimiser %d", who, opt_kind); return PTR_ERR_INVALID_ARG; }
    if (step < 1 || !state1 || (opt_kind == PTR_OPT_ADAM && !state2) || (loss_out && nq > 0 && !loss_q) || nq < 0) {
        set_error("%s: bad optimiser / loss arguments", who);
        return PTR_ERR_INVALID_ARG;
    }
    OptStep o{};
    o.kind = opt_kind; o.h1 = hyper1; o.h2 = hyper2; o.eps = eps; o.wd = weight_decay;
    o.param = params; o.s1 = state1; o.s2 = state2; o.loss_q = loss_q; o.nq = nq; o.loss_out = loss_out;
    o.wimg = reinterpret_cast<uint8_t *>(wimg); o.F = F; o.NL = NL;
    if (opt_kind == PTR_OPT_ADAM) {                                  // as ptr_adam_step
        o.lr = lr; o.bc1 = 1.0f - powf(hyper1, (float)step); o.bc2_sqrt = sqrtf(1.0f - powf(hyper2, (float)step));
    } else if (opt_kind == PTR_OPT_ADAGRAD) {                        // as ptr_adagrad_step: hyper1 = lr_decay
        o.lr = lr / (1.0f + (float)(step - 1) * hyper1);
    } else {
        o.lr = lr;                                                   // hyper1 = alpha
    }
    return mlp_backward_impl(who, X, params, acts, dpreds, R, F, NL, p_drop, seed, dz, ws, grad, stream, o);
}

// The optimiser step + the loss-slot sum as ONE launch on a gradient that is already reduced (data parallelism: backward -> flat gradient ->
// RCCL all-reduce -> this): reduce_partials_kernel over ONE "partial" — the gradient itself — so the arithmetic is ptr_mlp_backward_step's,
// bit for bit, and the data-parallel step is the single-device launch sequence + one kernel + one collective.
extern "C" int ptr_opt_step_loss(float *params, const float *grad, int64_t n, int opt_kind, float lr, float hyper1, float hyper2, float eps,
                                 float weight_decay, int step, float *state1, float *state2, const float *loss_q, int nq, float *loss_out,
                                 void *stream) {
    using namespace ptr;
    const char *who = "ptr_opt_step_loss";
    if (opt_kind < PTR_OPT_ADAM || opt_kind > PTR_OPT_RMSPROP) { set_error("%s: unknown optimiser %d", who, opt_kind); return PTR_ERR_INVALID_ARG; }
    if (n < 0 || step < 1 || nq < 0 || (n > 0 && (!params || !grad || !state1 || (opt_kind == PTR_OPT_ADAM && !state2))) || (loss_out && nq > 0 && !loss_q)) {
        set_error("%s: bad optimiser / loss arguments", who);
        return PTR_ERR_INVALID_ARG;
    }
    if (n == 0 && !loss_out) return 0;
    OptStep o{};
    o.kind = opt_kind; o.h1 = hyper1; o.h2 = hyper2; o.eps = eps; o.wd = weight_decay;
    o.param = params; o.s1 = state1; o.s2 = state2; o.loss_q = loss_q; o.nq = nq; o.loss_out = loss_out;
    if (opt_kind == PTR_OPT_ADAM) { o.lr = lr; o.bc1 = 1.0f - powf(hyper1, (float)step); o.bc2_sqrt = sqrtf(1.0f - powf(hyper2, (float)step)); }
    else if (opt_kind == PTR_OPT_ADAGRAD) o.lr = lr / (1.0f + (float)(step - 1) * hyper1);
    else o.lr = lr;
    const size_t nn = (size_t)n;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((nn + 63) / 64) + (loss_out ? 1 : 0)), dim3(1024), 0, as_stream(stream), grad, 1, 1, nn,
                       nn, nn, (float *)nullptr, o);
    return check_hip(hipGetLastError(), who);
}

int ptr::mlp_backward_impl(const char *who, const float *X, const float *params, const float *acts, const float *dpreds, int R, int F, int NL,
                                  float p_drop, uint64_t seed, float *dz, float *ws, float *grad, void *stream, const OptStep &opt) {
    using namespace ptr;
    if (int rc = check_mlp(who, R, F, NL, p_drop)) return rc;
    if (!X || !params || !acts || !dpreds || !ws || !grad) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    MlpArgs a{R, F, NL, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32)};
    if (R > 0 && bwd_fused_supported(F, NL, X, acts)) {
        // single pass: X and the stored activations are read once, dZ never leaves the chip (scorer_bwd.hip); dz is not touched
        // (bf16x6 formulation from PTR_BWD_X6's row threshold on, scorer_bwd_x6.hip; the fp32-MFMA kernel otherwise)
        if (int e = bwd_x6_supported(R, F, NL, X, acts) ? launch_bwd_x6(X, params, acts, dpreds, a, ws, st, who)
                                                        : launch_bwd_fused(X, params, acts, dpreds, a, ws, st, who)) return e;
        const int nb = bwd_fused_grid(R);
        const size_t NPf = n_params(NL, F);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((NPf + 63) / 64) + (opt.loss_out ? 1 : 0)), dim3(1024), 0, st, ws, nb, nb, NPf, NPf, NPf,
                           grad, opt);
        return check_hip(hipGetLastError(), who);
    }
    if (!dz) { set_error("%s: dz scratch is required for this configuration (ptr_mlp_backward_dz_floats)", who); return PTR_ERR_INVALID_ARG; }
    // 1. dZ chain (+ partial d w_out / d b_out)
    const int ncu = mlp_num_cus();
    const int nblk = dw_blocks_per_cu() * ncu;
    const size_t NP = n_params(NL, F);
    const bool wide = dz_wide() != 0;
    const int wpb = wide ? 16 : 8;
    const int ntiles = (R + 15) / 16;
    int grid_dz = ntiles < wpb * ncu ? (ntiles + wpb - 1) / wpb : ncu;
    if (grid_dz < 1) grid_dz = 1;
    // r4, NL = 2 / 3: ONE pass over the stored activations does the dZ chain AND every hidden-layer gradient (the fused kernel's TAIL form,
    // scorer_bwd.hip: no X image, no first-layer tiles) and leaves dZ of the first layer in dz[0]; only the first layer's dW below still
    // runs as a row-contraction kernel of its own.  Partials: the first layer's entries [0, off_W(1)) come from the dW kernels' grid, everything
    // behind them from the tail kernel's.
    const bool tail = R > 0 && bwd_tail_supported(NL, acts);
    if (tail) {
        // r5: on the bf16 instructions for three hidden layers (the pipelined kernel's TAIL form, scorer_bwd_x6.hip); fp32-MFMA otherwise / PTR_BWD_X6=0
        if (int e = bwd_x6_tail_supported(R, NL, acts) ? launch_bwd_x6_tail(params, acts, dpreds, a, ws, dz, st, who)
                                                       : launch_bwd_tail(params, acts, dpreds, a, ws, dz, st, who)) return e;
    } else {
        const size_t lds = dz_lds_floats(NL) * sizeof(float);
        auto go = [&](auto kern) -> int {
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3(grid_dz), dim3(wpb * 64), lds, st, params, acts, dpreds, a, dz, ws, NP, off_wout(NL, F));
            return check_hip(hipGetLastError(), who);
        };
        if (int e = wide ? go(mlp_bwd_dz_kernel<1, 1024>) : go(mlp_bwd_dz_kernel<1, 512>)) return e;
    }
    // 2. dW per layer (row contraction), every block writes its partial into ws[block][flat parameter layout]
    for (int l = 0; l < (tail ? 1 : NL); ++l) {
        const int K = l == 0 ? F : kH;
        const float *A = l == 0 ? X : acts + (size_t)(l - 1) * act_layer_floats(R);      // (tile-major: the !SITE0 kernels address it with act_off)
        const int lda = l == 0 ? F : kAL;
        const float *dZ = dz + (size_t)l * R * kAL;
        const int ntk = (K + 15) / 16;
        auto go = [&](auto kern, int nt_base) -> int {
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 0, st, A, lda, dZ, K, nt_base, a, ws, NP, off_W(l, F), off_b(l, F));
            return check_hip(hipGetLastError(), who);
        };
        auto go_lds = [&](auto kern, int ntw, int rb, int nt_base) -> int {
            if (int e0 = allow_lds(kern, dw_lds_bytes(ntw, rb))) return e0;
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), dw_lds_bytes(ntw, rb), st, A, lda, dZ, K, nt_base, a, ws, NP, off_W(l, F),
                               off_b(l, F));
            return check_hip(hipGetLastError(), who);
        };
        const bool aligned = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && dw_staged();
        int e = 0;
        if (l == 0) {   // in-feature tiles per wave and pass: 3 (192 columns) or 6 (384 columns)
            if (aligned) {
                if (ntk <= 12) e = dw_rb() == 32 ? go_lds(mlp_bwd_dw_lds_kernel<3, true, 32>, 3, 32, 0) : go_lds(mlp_bwd_dw_lds_kernel<3, true, 16>, 3, 16, 0);
                else if (dw_x6_supported(R, K, lda, A))   // wide inputs: the bf16x6 row contraction (scorer_dw_x6.hip), same partial layout
                    e = launch_dw_x6(A, lda, dZ, K, ntk, a, ws, NP, off_W(l, F), off_b(l, F), nblk, st, who);
                else for (int base = 0; base < ntk && !e; base += 24) e = go_lds(mlp_bwd_dw_lds_kernel<6, true, 16>, 6, 16, base);
            } else if (ntk <= 12) e = go(mlp_bwd_dw_kernel<3, true, 4>, 0);
            else if (ntk <= 24) e = go(mlp_bwd_dw_kernel<6, true, 4>, 0);
            else if (ntk <= 48) { e = go(mlp_bwd_dw_kernel<6, true, 4>, 0); if (!e) e = go(mlp_bwd_dw_kernel<6, true, 4>, 24); }
            else { set_error("%s: F=%d not supported by the dW kernel", who, F); return PTR_ERR_UNSUPPORTED; }
        } else {
            if (!aligned) e = go(mlp_bwd_dw_kernel<2, false, 4>, 0);
            else e = dw_rb() == 32 ? go_lds(mlp_bwd_dw_lds_kernel<2, false, 32>, 2, 32, 0) : go_lds(mlp_bwd_dw_lds_kernel<2, false, 16>, 2, 16, 0);
        }
        if (e) return e;
    }
    // 3. one deterministic reduction of all partials into the flat gradient
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((NP + 63) / 64) + (opt.loss_out ? 1 : 0)), dim3(1024), 0, st, ws, nblk,
                       tail ? bwd_fused_grid(R) : grid_dz, tail ? off_W(1, F) : off_wout(NL, F), NP, NP, grad, opt);
    return check_hip(hipGetLastError(), who);
}

namespace ptr {
// torch.optim.Adagrad (lr_decay, eps; initial accumulator 0) and torch.optim.RMSprop (alpha, eps; no momentum, not centered) on flat
// buffers — the other two optimisers the reference configures (ptranking/base/ranker.py:518-521), L2 weight decay added to the gradient
__global__ void __launch_bounds__(256)
adagrad_kernel(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ sum, size_t n, float clr, float eps, float wd) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float pi = p[i];
    const float gi = grad[i] + wd * pi;
    const float si = sum[i] + gi * gi;
    sum[i] = si;
    p[i] = pi - clr * (gi / (sqrtf(si) + eps));
}
__global__ void __launch_bounds__(256)
rmsprop_kernel(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ sq, size_t n, float lr, float alpha, float eps, float wd) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float pi = p[i];
    const float gi = grad[i] + wd * pi;
    const float si = alpha * sq[i] + (1.0f - alpha) * gi * gi;
    sq[i] = si;
    p[i] = pi - lr * (gi / (sqrtf(si) + eps));
}
}  // namespace ptr

extern "C" int ptr_adagrad_step(float *param, const float *grad, float *state_sum, int64_t n, float lr, float lr_decay, float eps,
                                float weight_decay, int step, void *stream) {
    using namespace ptr;
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !state_sum))) { set_error("ptr_adagrad_step: bad arguments"); return PTR_ERR_INVALID_ARG; }
    if (n == 0) return 0;
    const float clr = lr / (1.0f + (float)(step - 1) * lr_decay);
    hipLaunchKernelGGL(adagrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad, state_sum, (size_t)n, clr, eps,
                       weight_decay);
    return check_hip(hipGetLastError(), "ptr_adagrad_step");
}

extern "C" int ptr_rmsprop_step(float *param, const float *grad, float *square_avg, int64_t n, float lr, float alpha, float eps,
                                float weight_decay, void *stream) {
    using namespace ptr;
    if (n < 0 || (n > 0 && (!param || !grad || !square_avg))) { set_error("ptr_rmsprop_step: bad arguments"); return PTR_ERR_INVALID_ARG; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad, square_avg, (size_t)n, lr, alpha,
                       eps, weight_decay);
    return check_hip(hipGetLastError(), "ptr_rmsprop_step");
}

extern "C" int ptr_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, void *stream) {
    using namespace ptr;
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) { set_error("ptr_adam_step: bad arguments"); return PTR_ERR_INVALID_ARG; }
    if (n == 0) return 0;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       (size_t)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    return check_hip(hipGetLastError(), "ptr_adam_step");
}

extern "C" int ptr_mlp_dropout_mask(int R, int n_feat, int site, float p_drop, uint64_t seed, float *out, void *stream) {
    using namespace ptr;
    if (R < 0 || n_feat <= 0 || !out) { set_error("ptr_mlp_dropout_mask: bad arguments"); return PTR_ERR_INVALID_ARG; }
    MlpArgs a{R, n_feat, 1, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32)};
    const size_t n = (size_t)R * n_feat;
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), a, site, n_feat, out);
    return check_hip(hipGetLastError(), "ptr_mlp_dropout_mask");
}
