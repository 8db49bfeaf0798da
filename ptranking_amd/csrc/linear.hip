// Hand-written fp32-MFMA kernels for the tall-skinny linear layers of the scoring functions (R = B*L documents x K inputs x N
// outputs, K and N a few hundred at most): the listsf head / tail feed-forward stacks (ff_dims 128/256/512), its Q|K|V and fc
// projections, and the layer-wise pointsf path (any activation / batch norm).  They replace the library GEMMs the reference's
// nn.Linear modules execute (ptranking/base/utils.py:288-356, ptranking/base/list_ranker.py:176-254,303-350).
//
//   ptr_linear_forward          Y  = epilogue(X W^T + b)            "transposed world", as the fused pointsf forward (scorer.hip)
//   ptr_linear_backward_input   dX = gate(dY W)                     the same kernel on W^T (staged transposed into LDS)
//   ptr_linear_backward_weight  dW = dY^T X, db = sum_rows dY       "row-contraction world", LDS-staged slabs (as mlp_bwd_dw_lds)
//
// Forward / backward-input: Y^T[n][row] = W[n][k] * X^T[k][row] with v_mfma_f32_16x16x4_f32 (exact fp32).  A workgroup stages a
// tile of <= 128 outputs x all K of the weights into LDS once ([n][K + 4], read as one ds_read_b128 per four k-steps) and its 8
// waves walk row tiles of 32 documents; the B operand is a float4 per lane straight from X (16 documents x 64 B per wave-load,
// prefetched one 16-column super-step ahead), the accumulators (<= 8 x 2 tiles) leave as float4 stores.  Outputs beyond the LDS
// budget are covered by blockIdx.y (X is then re-read per output tile; it streams from L2 / HBM at far below the MFMA time).
// Epilogue: + bias, optional ReLU + dropout (counter-based, ptr_dropout.h — the stored value then is the NEXT layer's post-dropout
// input, `a > 0` encodes "ReLU active and kept"), or for the backward-input form the gate dX *= [gate > 0] * inv_keep.
//
// Backward-weight: documents are the MFMA k index; a workgroup owns a chunk of rows and one (<= 128 outputs) x (<= 256 inputs)
// block of dW, loads each 16-row slab of dY / X once with 16-byte loads into double-buffered LDS, and writes its partial block to
// `ws`; ptr_reduce_partials sums the chunks in a fixed order (deterministic).  db comes from the same dY slabs.
#include <stdlib.h>

#include "ptr_device.h"
#include "ptr_dropout.h"
#include "ptr_linear.h"

namespace ptr {


__host__ __device__ inline int lin_ldk(int K) { return (K + 3) / 4 * 4 + 4; }

// Y[r][n0 + n] = epi(sum_k X[r][k] * Wm[n0 + n][k] + bias[n0 + n]),  Wm = W ([N][K] row-major) or, TRANS, W^T with W [K][N].
// NW waves per workgroup, each walking tiles of 16 * RT documents.  r3: the pipeline of the fused pointsf forward (scorer.hip) —
//   * weight tile staged with batches of independent loads (a one-load-per-iteration loop waits out an L2 round trip per element; the
//     transposed form also divided by a run-time column count per element: ~10-18 us of prologue in a 200 us kernel);
//   * X two super-steps ahead through three register buffers that rotate BY NAME (a v_mov rotation reads the newest in-flight loads);
//   * the MT weight fragments of a super-step as ONE batch of LDS reads, MFMAs in groups of two tiles with the k-step outermost
//     (consecutive MFMAs write different accumulators);
//   * zero-padding selects only on a padded last super-step / the tail tile;
//   * 16 waves x 16-row tiles (4 waves per SIMD) by default, 8 waves x 32-row tiles for A/B (PTR_LIN_WIDE=0).
template <int MT, int RT, bool TRANS, bool VECX, int NW>
__global__ void __launch_bounds__(NW * 64)
linear_fwd_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                  const float *__restrict__ gate, LinArgs a, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NW * 64;
    constexpr int kAhead = 2;      // X super-steps in flight (three ahead through four buffers measured no different, hot or cold operands)
    const int K = a.K, N = a.N, R = a.R, ldk = lin_ldk(K);
    const int n0 = blockIdx.y * 16 * MT;
    float *Ws = smem;                                  // [16*MT][ldk]
    float *Bs = Ws + (size_t)16 * MT * ldk;            // [16*MT]
    const int tid = threadIdx.x;
    // ---- stage the weight tile and the bias; the zero padding (rows past N, columns K .. ldk-1) is written by a disjoint pass
    const int rows = min(16 * MT, N - n0);
    if constexpr (!TRANS) {
        if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
            const int k4 = K >> 2, n4 = rows * k4;
            constexpr int U = 4;
            for (int base = tid; base < n4; base += U * NT) {
                f32x4 v[U];
                int r[U], c[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = min(base + u * NT, n4 - 1);
                    r[u] = idx / k4; c[u] = idx - r[u] * k4;
                    v[u] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n0 + r[u]) * K + 4 * c[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + u * NT < n4) *reinterpret_cast<f32x4 *>(Ws + (size_t)r[u] * ldk + 4 * c[u]) = v[u];
            }
        } else {
            const int n = rows * K;
            constexpr int U = 4;
            for (int base = tid; base < n; base += U * NT) {
                float v[U];
                int r[U], c[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = min(base + u * NT, n - 1);
                    r[u] = idx / K; c[u] = idx - r[u] * K;
                    v[u] = W[(size_t)(n0 + r[u]) * K + c[u]];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + u * NT < n) Ws[(size_t)r[u] * ldk + c[u]] = v[u];
            }
        }
    } else {   // W is [K][N]: Ws[n][k] = W[k][n0 + n]; lane = output column (coalesced along n), wave = k class; no per-element division
        const int tn = tid & 63, tk = tid >> 6;
        constexpr int U = 4;
        for (int n = tn; n < rows; n += 64) {
            for (int k = tk; k < K; k += U * NW) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = W[(size_t)min(k + u * NW, K - 1) * N + n0 + n];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (k + u * NW < K) Ws[(size_t)n * ldk + k + u * NW] = v[u];
            }
        }
    }
    for (int idx = tid; idx < 16 * MT * ldk; idx += NT) {
        const int r = idx / ldk, c = idx - r * ldk;
        if (r >= rows || c >= K) Ws[idx] = 0.0f;
    }
    for (int i = tid; i < 16 * MT; i += NT) Bs[i] = (bias && n0 + i < N) ? bias[n0 + i] : 0.0f;
    __syncthreads();

    const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int rows_per_tile = 16 * RT;
    const int ntiles = (R + rows_per_tile - 1) / rows_per_tile;
    const int nS = (K + 15) >> 4;
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const bool vecy = ((a.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    const bool vecg = gate && ((a.ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(gate) & 15) == 0);

    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
        const int row0 = tile * rows_per_tile;
        const bool tile_full = row0 + rows_per_tile <= R;
        int row[RT];
        const float *xrow[RT];
        bool rok[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            row[rt] = row0 + 16 * rt + j;
            rok[rt] = row[rt] < R;
            xrow[rt] = X + (size_t)(rok[rt] ? row[rt] : R - 1) * a.ldx;
        }
        auto load_raw = [&](int S, f32x4 (&xb)[RT]) {      // raw loads from clamped addresses; masks applied after the MFMAs
            const int k0 = 16 * S + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (VECX) {
                    xb[rt] = *reinterpret_cast<const f32x4 *>(xrow[rt] + (k0 < K ? k0 : 0));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) xb[rt][c] = xrow[rt][k0 + c < K ? k0 + c : 0];
                }
            }
        };
        auto finish_x = [&](int S, f32x4 (&xb)[RT]) {      // uniform branch: only a padded super-step and the tail tile need the selects
            if (16 * S + 16 > K || !tile_full) {
                const int k0 = 16 * S + 4 * g;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xb[rt][c] *= ((k0 + c < K) && rok[rt]) ? 1.0f : 0.0f;
            }
        };
        f32x4 acc[MT][RT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(Bs + 16 * mt + 4 * g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
        }
        auto step = [&](int S, f32x4 (&cur)[RT], f32x4 (&nxt)[RT], f32x4 (&nn)[RT]) {
            load_raw(S + kAhead < nS ? S + kAhead : 0, nn);   // past the end: a valid address, never consumed
            __builtin_amdgcn_sched_barrier(0);                // the loads stay HERE (the scheduler sinks them towards their use)
            const int k0 = 16 * S + 4 * g;
            const float *wp = Ws + (size_t)j * ldk + (k0 < ldk - 3 ? k0 : 0);
            if constexpr (MT <= 8) {          // all fragments of the super-step as one batch of LDS reads
                f32x4 wa[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) wa[mt] = *reinterpret_cast<const f32x4 *>(wp + (size_t)16 * mt * ldk);
                if constexpr (RT > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m0 = 0; m0 < MT; m0 += 2)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int mt = m0; mt < (m0 + 2 < MT ? m0 + 2 : MT); ++mt)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[mt][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][c], cur[rt][c], acc[mt][rt], 0, 0, 0);
            } else {                          // 9..10 output tiles (X read once for N = 136): fragments per group of two, registers stay <= 128
#pragma unroll
                for (int m0 = 0; m0 < MT; m0 += 2) {
                    f32x4 wa[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) wa[i] = *reinterpret_cast<const f32x4 *>(wp + (size_t)16 * (m0 + i < MT ? m0 + i : m0) * ldk);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (m0 + i < MT) {
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt)
                                    acc[m0 + i][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i][c], cur[rt][c], acc[m0 + i][rt], 0, 0, 0);
                            }
                }
            }
            finish_x(S + 1, nxt);                             // S + 1 == nS: finishes values nobody reads
        };
        f32x4 xa[RT], xb[RT], xc[RT];
        load_raw(0, xa);
        load_raw(nS > 1 ? 1 : 0, xb);
        finish_x(0, xa);
        int S = 0;
        for (; S + 3 <= nS; S += 3) {
            step(S, xa, xb, xc);
            step(S + 1, xb, xc, xa);
            step(S + 2, xc, xa, xb);
        }
        if (nS - S == 1) {
            step(S, xa, xb, xc);
        } else if (nS - S == 2) {
            step(S, xa, xb, xc);
            step(S + 1, xb, xc, xa);
        }
        // ---- epilogue
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 h = acc[mt][rt];
                const int nb = n0 + 16 * mt + 4 * g;
                if (a.act == PTR_LINEAR_RELU_DROPOUT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
                    if (a.p_drop > 0.0f) {
                        uint32_t w0, w1;
                        drop_bits(a.seed_lo, a.seed_hi, a.site, row[rt], nb >> 2, w0, w1);
                        h = drop4(h, w0, w1, thr, inv_keep);
                    }
                } else if (a.act == PTR_LINEAR_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
                } else if (a.act == PTR_LINEAR_GATE) {
                    if (rok[rt]) {
                        const float *gp = gate + (size_t)row[rt] * a.ldg + nb;
                        if (vecg && nb + 3 < N) {                  // one 16-byte load per fragment (r2 issued four scalar loads)
                            const f32x4 gv = *reinterpret_cast<const f32x4 *>(gp);
#pragma unroll
                            for (int c = 0; c < 4; ++c) h[c] *= gv[c] > 0.0f ? inv_keep : 0.0f;
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float gv = nb + c < N ? gp[c] : 0.0f;
                                h[c] *= gv > 0.0f ? inv_keep : 0.0f;
                            }
                        }
                    }
                }
                if (rok[rt]) {
                    float *yp = Y + (size_t)row[rt] * a.ldy + nb;
                    if (vecy && nb + 3 < N) {
                        *reinterpret_cast<f32x4 *>(yp) = h;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (nb + c < N) yp[c] = h[c];
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------- backward: weights
// One workgroup = one chunk of rows x one block of dW: outputs [n0, n0 + 16*MTO) x inputs [k0, k0 + 64*NTW).  Wave w owns the
// in-feature tiles w, w+4, ...  ws[chunk][N*K + N]: partial dW (row-major [N][K]) followed by partial db.
template <int MTO, int NTW, int RB>
__global__ void __launch_bounds__(256)
linear_bwd_w_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ dY, int ldy, int R, int K, int N, int nkb,
                    float *__restrict__ ws, size_t ws_stride) {
    constexpr int WA4 = 16 * NTW;                  // float4 per X-slice row
    constexpr int LDA = 64 * NTW + 16;             // LDS row strides = 16 (mod 32)
    constexpr int WZ4 = 4 * MTO;                   // float4 per dY-slice row
    constexpr int LDZ = 16 * MTO + 16;
    constexpr int SA = (RB * WA4 + 255) / 256, SZ = (RB * WZ4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto zb = [&](int b) -> float * { return smem + b * (RB * LDZ); };
    auto ab = [&](int b) -> float * { return smem + 2 * RB * LDZ + b * (RB * LDA); };
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nb_ = blockIdx.y / nkb, kb_ = blockIdx.y - nb_ * nkb;
    const int n0 = nb_ * 16 * MTO, k0 = kb_ * 64 * NTW;
    const int chunk = ((R + gridDim.x - 1) / gridDim.x + RB - 1) / RB * RB;
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const bool vec = ((ldx & 3) == 0) && ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dY) & 15) == 0) && ((K & 3) == 0) && ((N & 3) == 0);
    int a_row[SA], a_col[SA], z_row[SZ], z_col[SZ];
    bool a_in[SA], z_in[SZ];
#pragma unroll
    for (int s_ = 0; s_ < SA; ++s_) {
        const int idx = s_ * 256 + tid;
        a_in[s_] = idx < RB * WA4;
        a_row[s_] = a_in[s_] ? idx / WA4 : 0;
        a_col[s_] = a_in[s_] ? 4 * (idx % WA4) : 0;
    }
#pragma unroll
    for (int s_ = 0; s_ < SZ; ++s_) {
        const int idx = s_ * 256 + tid;
        z_in[s_] = idx < RB * WZ4;
        z_row[s_] = z_in[s_] ? idx / WZ4 : 0;
        z_col[s_] = z_in[s_] ? 4 * (idx % WZ4) : 0;
    }
    f32x4 ra[SA], rz[SZ];
    auto gload = [&](int r0) {                      // raw loads (clamped addresses), masks applied in lstore
#pragma unroll
        for (int s_ = 0; s_ < SA; ++s_) {
            const int r = min(r0 + a_row[s_], r_end - 1), c = k0 + a_col[s_];
            const float *p = X + (size_t)r * ldx;
            if (vec) ra[s_] = *reinterpret_cast<const f32x4 *>(p + (c + 3 < K ? c : 0));
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[s_][e] = p[c + e < K ? c + e : 0];
        }
#pragma unroll
        for (int s_ = 0; s_ < SZ; ++s_) {
            const int r = min(r0 + z_row[s_], r_end - 1), c = n0 + z_col[s_];
            const float *p = dY + (size_t)r * ldy;
            if (vec) rz[s_] = *reinterpret_cast<const f32x4 *>(p + (c + 3 < N ? c : 0));
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) rz[s_][e] = p[c + e < N ? c + e : 0];
        }
    };
    auto lstore = [&](int buf, int r0) {
#pragma unroll
        for (int s_ = 0; s_ < SA; ++s_) {
            if (a_in[s_]) {
                const bool rok = r0 + a_row[s_] < r_end;
                f32x4 v = ra[s_];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (rok && k0 + a_col[s_] + e < K) ? 1.0f : 0.0f;
                *reinterpret_cast<f32x4 *>(ab(buf) + a_row[s_] * LDA + a_col[s_]) = v;
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < SZ; ++s_) {
            if (z_in[s_]) {
                const bool rok = r0 + z_row[s_] < r_end;
                f32x4 v = rz[s_];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (rok && n0 + z_col[s_] + e < N) ? 1.0f : 0.0f;
                *reinterpret_cast<f32x4 *>(zb(buf) + z_row[s_] * LDZ + z_col[s_]) = v;
            }
        }
    };
    f32x4 acc[NTW][MTO];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbv[MTO];
#pragma unroll
    for (int mt = 0; mt < MTO; ++mt) dbv[mt] = 0.0f;
    if (r_begin < r_end) {
        gload(r_begin);
        lstore(0, r_begin);
    }
    __syncthreads();
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += RB, buf ^= 1) {
        const bool more = r0 + RB < r_end;
        if (more) gload(r0 + RB);
        const float *zs = zb(buf), *as = ab(buf);
#pragma unroll
        for (int u = 0; u < RB / 4; ++u) {
            float av[MTO], bv[NTW];
#pragma unroll
            for (int mt = 0; mt < MTO; ++mt) { av[mt] = zs[(4 * u + g) * LDZ + 16 * mt + j]; dbv[mt] += av[mt]; }
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[t] = as[(4 * u + g) * LDA + 16 * (wave + 4 * t) + j];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int mt = 0; mt < MTO; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], acc[t][mt], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1, r0 + RB);
        __syncthreads();
    }
    float *out = ws + (size_t)blockIdx.x * ws_stride;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int k = k0 + 16 * (wave + 4 * t) + j;
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int o = n0 + 16 * mt + 4 * g + c;
                if (k < K && o < N) out[(size_t)o * K + k] = acc[t][mt][c];
            }
    }
    if (wave == 0 && kb_ == 0) {
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt) {
            float v = dbv[mt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int o = n0 + 16 * mt + j;
            if (g == 0 && o < N) out[(size_t)N * K + o] = v;
        }
    }
}

// out[i] = sum_b ws[b][i] in a fixed order: 16 elements x 16 partial lanes per workgroup (lane bl sums the partials bl, bl + 16, ... four
// at a time, the 16 lane sums are added in lane order) — one thread per element walking all the partials is a chain of nblk dependent
// additions behind nblk L2 round trips (40 us for 512 partials of a 100 x 100 gradient)
__global__ void __launch_bounds__(256)
reduce_chunks_kernel(const float *__restrict__ ws, int nblk, size_t stride, size_t n, float *__restrict__ dW, size_t n_w,
                     float *__restrict__ db) {
    constexpr int CL = 16, BL = 16;
    __shared__ float red[BL][CL + 1];
    const int cl = threadIdx.x & (CL - 1), bl = threadIdx.x / CL;
    const size_t i = (size_t)blockIdx.x * CL + cl;
    const bool on = i < n;
    auto at = [&](int b) { return b < nblk ? ws[(size_t)b * stride + i] : 0.0f; };
    float s = 0.0f;
    if (on)
        for (int b = bl; b < nblk; b += 4 * BL) s += (at(b) + at(b + BL)) + (at(b + 2 * BL) + at(b + 3 * BL));
    red[bl][cl] = s;
    __syncthreads();
    if (on && bl == 0) {
        float t = 0.0f;
        for (int k = 0; k < BL; ++k) t += red[k][cl];
        if (i < n_w) dW[i] = t;
        else if (db) db[i - n_w] = t;
    }
}

// ---------------------------------------------------------------------------------------------------- elementwise companions
// out[r][c] = x[r][c] * keep(seed, site, r, c) / (1 - p): nn.Dropout in front of a stack's first Linear (forward: x = features,
// backward: x = dLoss/d(dropped input) — the mask is recomputed from the counter-based generator, never stored).  C % 4 == 0.
__global__ void __launch_bounds__(256)
dropout_apply_kernel(const float *__restrict__ x, int ldx, int R, int C, float p_drop, uint32_t seed_lo, uint32_t seed_hi, int site,
                     float *__restrict__ out, int ldo) {
    const int c4 = C >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * c4) return;
    const int r = (int)(i / c4), c = (int)(i - (size_t)r * c4);
    f32x4 v = *reinterpret_cast<const f32x4 *>(x + (size_t)r * ldx + 4 * c);
    uint32_t w0, w1;
    drop_bits(seed_lo, seed_hi, site, r, c, w0, w1);
    v = drop4(v, w0, w1, drop_thr(p_drop), 1.0f / (1.0f - p_drop));
    *reinterpret_cast<f32x4 *>(out + (size_t)r * ldo + 4 * c) = v;
}

// out = dy * [y > 0]: the backward of a trailing ReLU whose output y was stored
__global__ void __launch_bounds__(256)
relu_gate_kernel(const float *__restrict__ dy, const float *__restrict__ y, size_t n, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

static int lin_num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// output tiles (of 16) per workgroup: as many as fit 150 KB of LDS next to K inputs, at most `cap` (10 for the 16-row form — 11 and 12 spill at 128 VGPRs —: N = 136 in ONE
// block — X is read once instead of twice, 9 tiles instead of 2 x 5; 8 for the 32-row form's registers), padding minimised
static void lin_tiling(int K, int N, int cap, int &MT, int &nblocks) {
    const int n16 = (N + 15) / 16;
    int mt_max = (int)((150 * 1024) / ((size_t)16 * lin_ldk(K) * sizeof(float) + 64));
    if (mt_max > cap) mt_max = cap;
    if (mt_max < 1) mt_max = 1;
    nblocks = (n16 + mt_max - 1) / mt_max;
    MT = (n16 + nblocks - 1) / nblocks;
}

template <bool TRANS>
static int launch_linear(const float *X, int ldx, const float *W, const float *bias, const float *gate, int ldg, int R, int K, int N,
                         int act, float p_drop, uint64_t seed, int site, float *Y, int ldy, hipStream_t st, const char *who) {
    if (R < 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) { set_error("%s: bad shape R=%d K=%d N=%d ldx=%d ldy=%d", who, R, K, N, ldx, ldy); return PTR_ERR_INVALID_ARG; }
    if (!X || !W || !Y || (act == PTR_LINEAR_GATE && !gate)) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) { set_error("%s: dropout p=%g out of [0,1)", who, (double)p_drop); return PTR_ERR_INVALID_ARG; }
    if (R == 0) return 0;
    {   // r6: the bf16x6 form of the same product (linear_x6.hip) where it serves the shape
        const LinArgs ax{R, K, N, ldx, ldy, ldg, act, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site};
        const int rc = launch_linear_x6(TRANS, X, W, bias, gate, ax, Y, lin_num_cus(), st, who);
        if (rc >= 0) return rc;
    }
    // 16 waves x 16-row tiles (4 waves per SIMD; <= 128 VGPRs) or, PTR_LIN_WIDE=0, the round-2 form 8 waves x 32-row tiles
    static const int wide = [] { const char *e = getenv("PTR_LIN_WIDE"); return e ? atoi(e) != 0 : 1; }();
    int MT, nby;
    lin_tiling(K, N, wide ? 10 : 8, MT, nby);
    const size_t lds = ((size_t)16 * MT * lin_ldk(K) + 16 * MT) * sizeof(float);
    if (lds > 160 * 1024) { set_error("%s: K=%d does not fit the LDS weight tile", who, K); return PTR_ERR_UNSUPPORTED; }
    LinArgs a{R, K, N, ldx, ldy, ldg, act, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site};
    const int nw = wide ? 16 : 8, rpt = wide ? 16 : 32;
    const int ntiles = (R + rpt - 1) / rpt;
    int gx = lin_num_cus() / nby;
    if (gx < 1) gx = 1;
    gx = ntiles < gx * nw ? (ntiles + nw - 1) / nw : gx;
    const bool vecx = ((ldx & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    auto go = [&](auto kern) -> int {
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(gx, nby), dim3(nw * 64), lds, st, X, W, bias, gate, a, Y);
        return check_hip(hipGetLastError(), who);
    };
#define LIN_CASE(M)                                                                                                     \
    case M:                                                                                                             \
        if (wide) return vecx ? go(linear_fwd_kernel<M, 1, TRANS, true, 16>) : go(linear_fwd_kernel<M, 1, TRANS, false, 16>);   \
        return vecx ? go(linear_fwd_kernel<M, 2, TRANS, true, 8>) : go(linear_fwd_kernel<M, 2, TRANS, false, 8>);
#define LIN_CASE_W(M) case M: return vecx ? go(linear_fwd_kernel<M, 1, TRANS, true, 16>) : go(linear_fwd_kernel<M, 1, TRANS, false, 16>);
    switch (MT) {
        LIN_CASE(1) LIN_CASE(2) LIN_CASE(3) LIN_CASE(4) LIN_CASE(5) LIN_CASE(6) LIN_CASE(7) LIN_CASE(8)
        LIN_CASE_W(9) LIN_CASE_W(10)
    }
#undef LIN_CASE_W
#undef LIN_CASE
    set_error("%s: internal tiling error", who);
    return PTR_ERR_UNSUPPORTED;
}

// backward-weight tiling: a block covers 16*MTO outputs x 64*NTW inputs (wave w owns the input tiles w, w+4, ...).  Both are chosen to
// minimise padding: K = 136 is 9 input tiles -> ONE block of NTW = 3 (a fixed 128-column block needed two, the second 94 % empty:
// 27 TFLOP/s measured); N = 136 is 9 output tiles -> two blocks of MTO = 5.  MTO * NTW <= 24 accumulator tiles (96 registers) per wave.
// linear_bw_x6.hip: the weight gradient on the bf16 instructions when one side of the product is narrow (plan 0 = not served)
int lin_bw_x6_plan(int R, int K, int N, int ldx, int ldy, const void *X, const void *dY);
int lin_bw_x6_chunks(int R, int K, int N, int plan);
int launch_lin_bw_x6(int plan, const float *X, int ldx, const float *dY, int ldy, int R, int K, int N, float *ws, int chunks, hipStream_t st, const char *who);

static void bw_tiling(int K, int N, int &MTO, int &NTW) {
    const int n_out = (N + 15) / 16, n_in = (K + 15) / 16;
    const int nbn = (n_out + 7) / 8;               // (r3: 9 tiles per block — N = 136 in one block, 172 VGPRs — measured SLOWER: 191 -> 216 us)
    MTO = (n_out + nbn - 1) / nbn;
    int ntw_max = 24 / MTO;
    if (ntw_max > 4) ntw_max = 4;
    const int nbk = (n_in + 4 * ntw_max - 1) / (4 * ntw_max);
    NTW = (n_in + 4 * nbk - 1) / (4 * nbk);
}
static int bw_chunks(int R, int K, int N) {
    int MTO, NTW;
    bw_tiling(K, N, MTO, NTW);
    const int blocks_y = ((N + 16 * MTO - 1) / (16 * MTO)) * ((K + 64 * NTW - 1) / (64 * NTW));
    int chunks = (2 * lin_num_cus() + blocks_y - 1) / blocks_y;
    const int max_chunks = (R + 255) / 256;        // at least 256 rows per chunk
    if (chunks > max_chunks) chunks = max_chunks;
    return chunks < 1 ? 1 : chunks;
}

}  // namespace ptr

extern "C" int ptr_linear_forward(const float *X, int ldx, const float *W, const float *bias, int R, int K, int N, int act, float p_drop,
                                  uint64_t seed, int site, float *Y, int ldy, void *stream) {
    if (act != PTR_LINEAR_NONE && act != PTR_LINEAR_RELU && act != PTR_LINEAR_RELU_DROPOUT) {
        ptr::set_error("ptr_linear_forward: unknown epilogue %d", act);
        return PTR_ERR_INVALID_ARG;
    }
    return ptr::launch_linear<false>(X, ldx, W, bias, nullptr, 0, R, K, N, act, act == PTR_LINEAR_RELU_DROPOUT ? p_drop : 0.0f, seed, site, Y, ldy,
                                     ptr::as_stream(stream), "ptr_linear_forward");
}

// dX[R][K] = (dY[R][N] W[N][K]) (* [gate > 0] / (1 - p_drop) when gate != NULL)
extern "C" int ptr_linear_backward_input(const float *dY, int ldy, const float *W, int R, int K, int N, const float *gate, int ldg,
                                         float p_drop, float *dX, int ldx, void *stream) {
    return ptr::launch_linear<true>(dY, ldy, W, nullptr, gate, ldg, R, N, K, gate ? PTR_LINEAR_GATE : PTR_LINEAR_NONE, p_drop, 0, 0, dX, ldx,
                                    ptr::as_stream(stream), "ptr_linear_backward_input");
}

extern "C" size_t ptr_linear_backward_weight_ws_floats(int R, int K, int N) {
    // the larger of the two kernels' needs (the bf16x6 form, linear_bw_x6.hip, runs one or two row chunks per CU)
    const int c32 = ptr::bw_chunks(R, K, N), c6 = (K <= 140 || N <= 144) ? ptr::lin_bw_x6_chunks(R, K, N, 0) : 0;
    return (size_t)(c32 > c6 ? c32 : c6) * ((size_t)N * K + N);
}

// dW[N][K] = dY^T X,  db[N] = column sums of dY (db may be NULL)
extern "C" int ptr_linear_backward_weight(const float *X, int ldx, const float *dY, int ldy, int R, int K, int N, float *ws, float *dW,
                                          float *db, void *stream) {
    using namespace ptr;
    const char *who = "ptr_linear_backward_weight";
    if (R < 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) { set_error("%s: bad shape", who); return PTR_ERR_INVALID_ARG; }
    if (!X || !dY || !ws || !dW) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    const size_t nw = (size_t)N * K, n = nw + N;
    if (R == 0) {
        if (int e = check_hip(hipMemsetAsync(dW, 0, nw * sizeof(float), st), who)) return e;
        return db ? check_hip(hipMemsetAsync(db, 0, N * sizeof(float), st), who) : 0;
    }
    if (const int plan = lin_bw_x6_plan(R, K, N, ldx, ldy, X, dY)) {
        const int chunks6 = lin_bw_x6_chunks(R, K, N, plan);
        if (int e = launch_lin_bw_x6(plan, X, ldx, dY, ldy, R, K, N, ws, chunks6, st, who)) return e;
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, ws, chunks6, n, n, dW, nw, db);
        return check_hip(hipGetLastError(), who);
    }
    int MTO, NTW;
    bw_tiling(K, N, MTO, NTW);
    const int nnb = (N + 16 * MTO - 1) / (16 * MTO), nkb = (K + 64 * NTW - 1) / (64 * NTW);
    const int chunks = bw_chunks(R, K, N);
#ifndef LIN_BW_RB
#define LIN_BW_RB 16
#endif
    constexpr int RB = LIN_BW_RB;
    auto go = [&](auto kern, int mto, int ntw) -> int {
        const size_t lds = (size_t)(2 * RB * (16 * mto + 16) + 2 * RB * (64 * ntw + 16)) * sizeof(float);
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(chunks, nnb * nkb), dim3(256), lds, st, X, ldx, dY, ldy, R, K, N, nkb, ws, n);
        return check_hip(hipGetLastError(), who);
    };
    int e = PTR_ERR_UNSUPPORTED;
#define BW_CASE(M, T) if (MTO == M && NTW == T) e = go(linear_bwd_w_kernel<M, T, RB>, M, T);
#define BW_ROW(M) BW_CASE(M, 1) BW_CASE(M, 2) BW_CASE(M, 3) BW_CASE(M, 4)
    BW_ROW(1) BW_ROW(2) BW_ROW(3) BW_ROW(4) BW_ROW(5) BW_ROW(6)
    BW_CASE(7, 1) BW_CASE(7, 2) BW_CASE(7, 3) BW_CASE(8, 1) BW_CASE(8, 2) BW_CASE(8, 3)
#undef BW_ROW
#undef BW_CASE
    if (e) return e;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, ws, chunks, n, n, dW, nw, db);
    return check_hip(hipGetLastError(), who);
}

extern "C" int ptr_dropout_apply(const float *x, int ldx, int R, int C, float p_drop, uint64_t seed, int site, float *out, int ldo,
                                 void *stream) {
    using namespace ptr;
    const char *who = "ptr_dropout_apply";
    if (R < 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3) || ldx < C || ldo < C || !(p_drop >= 0.0f && p_drop < 1.0f)) { set_error("%s: bad arguments (C, ldx, ldo must be multiples of 4)", who); return PTR_ERR_INVALID_ARG; }
    if (R > 0 && (!x || !out || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))) { set_error("%s: NULL / unaligned pointer", who); return PTR_ERR_INVALID_ARG; }
    if (R == 0) return 0;
    const size_t n = (size_t)R * (C >> 2);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, ldx, R, C, p_drop,
                       (uint32_t)seed, (uint32_t)(seed >> 32), site, out, ldo);
    return check_hip(hipGetLastError(), who);
}

extern "C" int ptr_relu_gate(const float *dy, const float *y, int64_t n, float *out, void *stream) {
    using namespace ptr;
    if (n < 0 || (n > 0 && (!dy || !y || !out))) { set_error("ptr_relu_gate: bad arguments"); return PTR_ERR_INVALID_ARG; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), dy, y, (size_t)n, out);
    return check_hip(hipGetLastError(), "ptr_relu_gate");
}
