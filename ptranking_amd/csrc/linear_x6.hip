// bf16x6 forward / backward-input of the tall-skinny linear layers (r6): Y[R][N] = epi(X[R][K] Wm^T + bias) with every fp32 product as six
// v_mfma_f32_16x16x32_bf16 products on exactly split operands (fp32 accumulation, fp32 results — the arithmetic of scorer_x6.hip) instead of
// v_mfma_f32_16x16x4_f32: the matrix time of a layer drops by 2.6x and the 136 / 100 / 128-wide layers of the listsf encoder and of the
// layer-wise pointsf stack become bound by their X and Y streams.  Serves 16-byte aligned rows of any K (whole-K weight image in LDS up to 256 inputs, K in
// chunks of 128 beyond); narrow outputs and small batches stay on linear.hip's fp32-MFMA kernel (launch_linear_x6 returns < 0).
//
// Reference: the nn.Linear modules of ptranking/base/utils.py:288-356 (`get_stacked_FFNet`) and ptranking/base/list_ranker.py:176-254,303-350
// (encoder projections, head / tail feed-forward stacks) and their autograd backward w.r.t. the input.
//
// "Transposed world", like linear.hip: acc tile [16 out-features][16 documents].  A workgroup (8 waves) splits its block of <= 144 out-features x
// all K of the weights ONCE into three bf16 planes in LDS, laid out FRAGMENT-major like the fused scorer's weight image ([k-step][plane][tile][lane
// group g][row j][16 bytes]: a fragment read is one contiguous, conflict-free KB at a compile-time offset from the k-step's base) and walks tiles of 32
// documents per wave: the X fragment of a k-step (lane (document j, k group g): two 16-byte loads, k = 4 g .. and 16 + 4 g .., two k-steps ahead) is split in registers (44 VALU per 32 k, amortised over all out-feature tiles), a weight
// fragment is one ds_read_b128 per plane per (tile, k-step) shared by the two document tiles.  K = 32 m + r with r <= 16 ends with ONE 16-deep
// k-step (v_mfma_f32_16x16x16_bf16 on 8-byte fragments): K = 136 is 4.5 k-steps, K = 100 is 3.5.
// Epilogue = linear.hip's: + bias (accumulator init), ReLU (+ counter-based dropout), or the backward-input gate.
#include <stdlib.h>

#include "ptr_device.h"
#include "ptr_dropout.h"
#include "ptr_linear.h"

namespace ptr {
namespace {

using lx_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using lx_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using lx_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using lx_u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using lx_i16x4 = __attribute__((ext_vector_type(4))) short;
using lx_f32x2 = __attribute__((ext_vector_type(2))) float;
union LxFrag { lx_bf16x8 v; lx_u32x4 q; uint32_t u[4]; };
using lx_lds_u32x4 = __attribute__((address_space(3))) lx_u32x4;
using lx_lds_u32x2 = __attribute__((address_space(3))) lx_u32x2;
__device__ __forceinline__ uint32_t lx_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ lx_u32x4 lx_lds16(uint32_t addr) { return *reinterpret_cast<const lx_lds_u32x4 *>((uintptr_t)addr); }
__device__ __forceinline__ lx_u32x2 lx_lds8(uint32_t addr) { return *reinterpret_cast<const lx_lds_u32x2 *>((uintptr_t)addr); }

constexpr int kLxNW = 8;          // waves per workgroup (two per SIMD: 256 registers each)
constexpr int kLxRT = 2;          // 16-document tiles per wave
constexpr int kLxNT = kLxNW * 64;

__host__ __device__ constexpr int lx_steps(int K) { return (((K + 15) & ~15) + 31) / 32; }      // k-steps incl. a 16-deep tail

__device__ __forceinline__ uint32_t lx_cvt_pk(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(lx_f32x2{x0, x1}, lx_bf16x2)); }
// round-to-nearest split of two fp32 values into one dword of each plane (scorer_x6.hip split_pack2: x = p1 + p2 + p3 exactly)
__device__ __forceinline__ void lx_split2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = lx_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = lx_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = lx_cvt_pk(s0, s1);
}
__device__ __forceinline__ void lx_split4(const f32x4 v, LxFrag (&f)[3], int d) {
    lx_split2(v[0], v[1], f[0].u[d], f[1].u[d], f[2].u[d]);
    lx_split2(v[2], v[3], f[0].u[d + 1], f[1].u[d + 1], f[2].u[d + 1]);
}
// four consecutive weights of one row -> 8 bytes of each plane image
__device__ __forceinline__ void lx_put4(uint8_t *dst, int plane, const f32x4 v) {
    uint32_t a[3], b[3];
    lx_split2(v[0], v[1], a[0], a[1], a[2]);
    lx_split2(v[2], v[3], b[0], b[1], b[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<lx_u32x2 *>(dst + (size_t)p * plane) = lx_u32x2{a[p], b[p]};
}

// the six products of one fp32 product, small terms first (scorer_x6.hip mma_tile)
constexpr int kLxA[6] = {0, 1, 2, 0, 1, 0}, kLxB[6] = {2, 1, 0, 1, 0, 0};

// ---- the weight block [n0, n0 + 16 MT) x K as three bf16 plane images in LDS, zero padded (rows past N, k >= K), followed by the bias; batches of four
// independent loads per thread.  k slots of a 32-deep k-step: lane (j, g) owns k = 4 g .. 4 g + 3 (fragment bytes 0..7) and 16 + 4 g .. 16 + 4 g + 3 (bytes
// 8..15) — any assignment works as long as both operands use it, and with this one a wave's X load covers 64 contiguous bytes per document.
// Element (row r, k): s SL + p PL + (r / 16) 1024 + ((k % 16) / 4) 256 + (r % 16) 16 + ((k % 32) / 16) 8 + 2 (k % 4); in the 16-deep tail step lane (j, g) owns
// k = 4 g .. 4 g + 3: nfull SL + p PL + (r / 16) 1024 + ((k % 32) / 4) 128 + (r % 16) 8 + 2 (k % 4).
// c0 (chunked form): the image holds the k-steps from k = 4 c0 on; nfull / nsteps count the steps of the image.
template <int MT, bool TRANS>
__device__ __forceinline__ void lx_stage(const float *__restrict__ W, int K, int N, int n0, int nfull, int nsteps, uint8_t *smem, int c0 = 0) {
    constexpr int PL = MT * 1024, SL = 3 * PL;
    const int rows = min(16 * MT, N - n0), tid = threadIdx.x;
    const int k4 = nsteps * 8, n4 = 16 * MT * k4;            // 16-byte groups of four k per row (the tail step: only its first four hold data)
    constexpr int U = 4;
    for (int base = tid; base < n4; base += U * kLxNT) {
        f32x4 v[U];
        int r[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = min(base + u * kLxNT, n4 - 1);
            if constexpr (!TRANS) { r[u] = idx / k4; c[u] = idx - r[u] * k4; }            // W [N][K]: a thread reads 16 bytes of one row (K % 4 == 0)
            else { c[u] = idx / (16 * MT); r[u] = idx - c[u] * (16 * MT); }               // W [K][N]: lanes along the out-features (coalesced), four k per thread
            const int kc = 4 * (c0 + c[u]);                                               // first of the four k of this group
            const bool in = r[u] < rows && kc < K;
            if constexpr (!TRANS) {
                v[u] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n0 + (in ? r[u] : 0)) * K + (in ? kc : 0));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[u][i] = W[(size_t)(in ? kc + i : 0) * N + n0 + (in ? r[u] : 0)];
            }
            if (!in) v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * kLxNT >= n4) continue;
            const int s_ = c[u] >> 3, cc = c[u] & 7;                                     // k-step, four-k group inside it
            const int tl = r[u] >> 4, rj = r[u] & 15;
            if (s_ < nfull) lx_put4(smem + (size_t)s_ * SL + tl * 1024 + (cc & 3) * 256 + rj * 16 + (cc >> 2) * 8, PL, v[u]);
            else if (cc < 4) lx_put4(smem + (size_t)s_ * SL + tl * 1024 + cc * 128 + rj * 8, PL, v[u]);
        }
    }
}
template <int MT>
__device__ __forceinline__ void lx_stage_bias(const float *__restrict__ bias, int N, int n0, float *Bs) {
    for (int i = threadIdx.x; i < 16 * MT; i += kLxNT) Bs[i] = (bias && n0 + i < N) ? bias[n0 + i] : 0.0f;
}

// ---- one 32-deep k-step of a wave: X fragments (already masked) split into planes, weight fragments one tile ahead through two named buffers.  The
// barriers keep the scheduler from hoisting every fragment read of the k-step to its top (MT = 9: 108 fragment registers on top of 72 accumulators) and from
// sinking the reads of the next tile behind this tile's MFMAs (to the next tile's doorstep).
template <int MT>
__device__ __forceinline__ void lx_kstep(uint32_t ab, const f32x4 (&x)[kLxRT][2], f32x4 (&acc)[MT][kLxRT]) {
    constexpr int PL = MT * 1024, RT = kLxRT;
    asm volatile("" : "+v"(ab));                  // ab: an LDS byte address.  One base register per k-step, every fragment at an immediate offset (<= 27 KB) from it: left to the compiler, the
                                                  // bases of all k-steps plus the > 64 KB offsets that do not fit the immediate field are hoisted into ~45 registers
    LxFrag bf[RT][3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { lx_split4(x[rt][0], bf[rt], 0); lx_split4(x[rt][1], bf[rt], 2); }
    LxFrag af[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) af[0][p].q = lx_lds16(ab + p * PL);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt + 1 < MT) {
#pragma unroll
            for (int p = 0; p < 3; ++p) af[(mt + 1) & 1][p].q = lx_lds16(ab + p * PL + (mt + 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[mt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt & 1][kLxA[q]].v, bf[rt][kLxB[q]].v, acc[mt][rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the 16-deep tail k-step (v_mfma_f32_16x16x16_bf16 on 8-byte fragments; x[rt][0] only)
template <int MT>
__device__ __forceinline__ void lx_ktail(uint32_t ab, const f32x4 (&x)[kLxRT][2], f32x4 (&acc)[MT][kLxRT]) {
    constexpr int PL = MT * 1024, RT = kLxRT;
    asm volatile("" : "+v"(ab));
    LxFrag bf[RT][3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) lx_split4(x[rt][0], bf[rt], 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        lx_u32x2 af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = lx_lds8(ab + p * PL + mt * 1024);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[mt][rt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(lx_i16x4, af[kLxA[q]]),
                                                                         __builtin_bit_cast(lx_i16x4, lx_u32x2{bf[rt][kLxB[q]].u[0], bf[rt][kLxB[q]].u[1]}),
                                                                         acc[mt][rt], 0, 0, 0);
    }
}

// ---- epilogue of one tile (linear.hip's arithmetic): ReLU (+ counter-based dropout) or the backward-input gate, 16-byte stores.  BRANCH-FREE memory
// operations: stores (and the gate loads) go through raw buffer resources with a 32-bit byte offset, and a lane that must not store (row past R, columns
// past N) uses an offset outside the buffer — the hardware drops it.  With `if (row < R) store` the stores sit in conditional blocks, the compiler cannot
// count how many of them are younger than an X load, and its s_waitcnt for that load then waits for (nearly) all outstanding operations — the epilogue's
// stores included: every tile waited for the write acknowledgements of the previous one.
using lx_rsrc = __amdgpu_buffer_rsrc_t;
constexpr uint32_t kLxOob = 0x80000000u;       // launch_linear_x6 serves outputs (and gates) below 2 GB
struct LxEpi {
    lx_rsrc Y, gate;
    int n0, N, ldy, ldg, act, site;
    float p_drop, inv_keep;
    uint32_t seed_lo, seed_hi, thr;
};
template <int MT, bool GATE>
__device__ __forceinline__ void lx_store(const LxEpi &e, const f32x4 (&acc)[MT][kLxRT], const int (&row)[kLxRT], const bool (&rok)[kLxRT], int g) {
    asm volatile("" : "+v"(g));                   // opaque per tile: the per-column offsets and dropout keys below are cheap to recompute, and hoisted out of the
                                                  // tile loop they cost the registers the accumulators need (16 spilled, their reloads drain the memory pipeline)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rt = 0; rt < kLxRT; ++rt) {
            f32x4 h = acc[mt][rt];
            const int nb = e.n0 + 16 * mt + 4 * g;
            const bool ok = rok[rt] && nb < e.N;                                  // N % 4 == 0: the four columns are in or out together
            if constexpr (GATE) {
                const uint32_t goff = ok ? (uint32_t)(row[rt] * e.ldg + nb) * 4u : kLxOob;     // out of range reads zeros
                const f32x4 gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(e.gate, (int)goff, 0, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] *= gv[c] > 0.0f ? e.inv_keep : 0.0f;
            } else if (e.act == PTR_LINEAR_RELU_DROPOUT) {
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
                if (e.p_drop > 0.0f) {
                    uint32_t w0, w1;
                    drop_bits(e.seed_lo, e.seed_hi, e.site, row[rt], nb >> 2, w0, w1);
                    h = drop4(h, w0, w1, e.thr, e.inv_keep);
                }
            } else if (e.act == PTR_LINEAR_RELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
            }
            const uint32_t yoff = ok ? (uint32_t)(row[rt] * e.ldy + nb) * 4u : kLxOob;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lx_u32x4, h), e.Y, (int)yoff, 0, 0);
        }
}
__device__ __forceinline__ LxEpi lx_epi(const LinArgs &a, const float *gate, float *Y, int n0) {
    LxEpi e;
    e.Y = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((size_t)a.R * a.ldy * 4), 0x00020000);
    e.gate = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gate ? gate : Y), 0, gate ? (int)((size_t)a.R * a.ldg * 4) : 0, 0x00020000);
    e.n0 = n0; e.N = a.N; e.ldy = a.ldy; e.ldg = a.ldg; e.act = a.act; e.site = a.site;
    e.p_drop = a.p_drop; e.inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    e.seed_lo = a.seed_lo; e.seed_hi = a.seed_hi; e.thr = drop_thr(a.p_drop);
    return e;
}

// =====================================================================================================================================================
// The form for K <= 144 (NFULL 32-deep k-steps + TAIL, compile time): a tile's WHOLE X block (<= 72 registers) is requested one tile ahead.  A wave's
// vector-memory operations retire in order: a load issued behind the 18 stores of an epilogue waits for their write acknowledgements, and the stores of
// 2048 waves drain at the HBM write rate (r6 ablation at 136 -> 136, 262 144 rows: 37 us without the stores, 95 us with them, the k-steps' loads queued behind
// them).  Here k-step s of tile t, once it has split its X fragment, requests the fragment of k-step s of tile t + 1 into the same registers — every load of
// tile t + 1 is in front of tile t's stores, and tile t + 1 multiplies while they drain.
template <int MT, bool TRANS, bool GATE, int NFULL, bool TAIL>
__global__ void __launch_bounds__(kLxNT)
linear_fwd_x6t_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ gate, LinArgs a,
                      float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_lx[];
    constexpr int RT = kLxRT, NS = NFULL + (TAIL ? 1 : 0);
    constexpr int PL = MT * 1024, SL = 3 * PL;
    const int K = a.K, R = a.R;
    const int n0 = blockIdx.y * 16 * MT;
    lx_stage<MT, TRANS>(W, K, a.N, n0, NFULL, NS, smem_lx);
    lx_stage_bias<MT>(bias, a.N, n0, reinterpret_cast<float *>(smem_lx + (size_t)NS * SL));
    __syncthreads();
    const float *Bs = reinterpret_cast<const float *>(smem_lx + (size_t)NS * SL);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int ntiles = (R + 16 * RT - 1) / (16 * RT);
    const LxEpi epi = lx_epi(a, gate, Y, n0);
    const uint32_t afrag = lx_lds_addr(smem_lx) + g * 256 + j * 16;
    const uint32_t afrag_t = lx_lds_addr(smem_lx) + NFULL * SL + g * 128 + j * 8;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // column offsets (floats) of the lane's two 16-byte pieces of every k-step, clamped to column 0 past K (selected away later)
    int ka[NS], kb[NS];
    bool oka[NS], okb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k0 = 32 * s + 4 * g;
        oka[s] = k0 < K; okb[s] = s < NFULL && k0 + 16 < K;
        ka[s] = oka[s] ? k0 : 0; kb[s] = okb[s] ? k0 + 16 : 0;
    }
    const int tile_stride = gridDim.x * kLxNW;
    int tile = blockIdx.x * kLxNW + wave;
    int row[RT], row_n[RT];
    bool rok[RT], rok_n[RT];
    const float *xrow_n[RT];
    auto next_rows = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            row_n[rt] = t * 16 * RT + 16 * rt + j;
            rok_n[rt] = row_n[rt] < R;
            xrow_n[rt] = X + (size_t)(rok_n[rt] ? row_n[rt] : R - 1) * a.ldx;
        }
    };
    f32x4 xs[NS][RT][2];
    // The X loads are HAND-COUNTED (inline asm the compiler's s_waitcnt insertion does not see): it would wait for a loop-carried load with vmcnt(0) — for the 18
    // stores issued behind it as well — because at the loop header it cannot tell how many younger operations are outstanding.  Here every operation of the
    // tile loop is unconditional, so the number issued behind the fragment of k-step s is the same at every use: the 2 RT loads of each of the other NS - 1
    // k-steps and the PREVIOUS tile's MT RT stores (+ MT RT gate loads): kBehind.  The kernel has no other vector-memory operation in the loop (0
    // spills: a scratch access would shift the count; tests/test_linear_gpu.py would see it).
    constexpr int kBehind = 2 * RT * (NS - 1) + MT * RT * (GATE ? 2 : 1);
    static_assert(kBehind < 64, "vmcnt is a 6-bit counter");
    auto request = [&](int s) __attribute__((always_inline)) {     // k-step s of the tile described by xrow_n
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xs[s][rt][0]) : "v"(xrow_n[rt] + ka[s]) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xs[s][rt][1]) : "v"(xrow_n[rt] + kb[s]) : "memory");
        }
    };
    next_rows(tile < ntiles ? tile : 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) request(s);
    // the first tile has no stores behind its loads: one full wait (kBehind would let its last k-steps through early)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; tile < ntiles; tile += tile_stride) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { row[rt] = row_n[rt]; rok[rt] = rok_n[rt]; }
        next_rows(tile + tile_stride < ntiles ? tile + tile_stride : tile);
        f32x4 acc[MT][RT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(Bs + 16 * mt + 4 * g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 x[RT][2];
            static_assert(RT == 2, "the wait below names the four registers of a k-step");
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xs[s][0][0]), "+v"(xs[s][0][1]), "+v"(xs[s][1][0]), "+v"(xs[s][1][1]) : "n"(kBehind) : "memory");
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                x[rt][0] = (rok[rt] && oka[s]) ? xs[s][rt][0] : zero4;
                x[rt][1] = (rok[rt] && okb[s]) ? xs[s][rt][1] : zero4;
            }
            request(s);                                       // the same k-step of the NEXT tile, in front of this tile's stores
            __builtin_amdgcn_sched_barrier(0);
            if (s < NFULL) lx_kstep<MT>(afrag + s * SL, x, acc);
            else lx_ktail<MT>(afrag_t, x, acc);
        }
        lx_store<MT, GATE>(epi, acc, row, rok, g);
    }
}

// =====================================================================================================================================================
// The general form (K <= 256): X two k-steps ahead through three buffers rotating BY NAME (a v_mov rotation would read the newest in-flight loads); the next
// tile's first two k-steps are requested in front of the epilogue's stores.
template <int MT, bool TRANS, bool GATE>
__global__ void __launch_bounds__(kLxNT)
linear_fwd_x6_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ gate, LinArgs a,
                     float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_lx[];
    constexpr int RT = kLxRT;
    constexpr int PL = MT * 1024, SL = 3 * PL;                // bytes of one plane / of one k-step of the weight image
    const int K = a.K, R = a.R;
    const int K16 = (K + 15) & ~15, nfull = K16 >> 5;
    const bool tail = (K16 & 31) != 0;                        // one 16-deep k-step behind the nfull 32-deep ones
    const int nsteps = nfull + (tail ? 1 : 0);
    const int n0 = blockIdx.y * 16 * MT;
    lx_stage<MT, TRANS>(W, K, a.N, n0, nfull, nsteps, smem_lx);
    lx_stage_bias<MT>(bias, a.N, n0, reinterpret_cast<float *>(smem_lx + (size_t)nsteps * SL));
    __syncthreads();
    const float *Bs = reinterpret_cast<const float *>(smem_lx + (size_t)nsteps * SL);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int ntiles = (R + 16 * RT - 1) / (16 * RT);
    const LxEpi epi = lx_epi(a, gate, Y, n0);
    const uint32_t afrag = lx_lds_addr(smem_lx) + g * 256 + j * 16;                 // + S SL + p PL + mt 1024: ds_read_b128
    const uint32_t afrag_t = lx_lds_addr(smem_lx) + nfull * SL + g * 128 + j * 8;   // the tail's 8-byte fragments
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    // (row, rok, xrow) describe the tile whose X fragments are being loaded / multiplied; the tile loop hands them over one tile early (below)
    int row[RT];
    const float *xrow[RT];
    bool rok[RT];
    auto set_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            row[rt] = t * 16 * RT + 16 * rt + j;
            rok[rt] = row[rt] < R;
            xrow[rt] = X + (size_t)(rok[rt] ? row[rt] : R - 1) * a.ldx;
        }
    };
    // raw loads from clamped addresses (a k-step past the end reads column 0 and is never consumed); the selects come later.  Both halves are always
    // loaded and always selected — branch-free on purpose: conditional stores into the buffer arrays make the compiler merge them through pointer phis,
    // the arrays then live in scratch memory and every k-step waits for its own prefetch (first build: 127 us where this one takes 85)
    auto load_raw = [&](int S, f32x4 (&xb)[RT][2]) __attribute__((always_inline)) {
        const bool full = S < nfull, last = tail && S == nfull;
        const int k0 = (full ? 32 * S : 32 * nfull) + 4 * g;
        const int ka = ((full || last) && k0 < K) ? k0 : 0, kb = (full && k0 + 16 < K) ? k0 + 16 : 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            xb[rt][0] = *reinterpret_cast<const f32x4 *>(xrow[rt] + ka);
            xb[rt][1] = *reinterpret_cast<const f32x4 *>(xrow[rt] + kb);
        }
    };
    auto finish_x = [&](int S, f32x4 (&xb)[RT][2]) __attribute__((always_inline)) {
        const bool full = S < nfull;
        const int k0 = (full ? 32 * S : 32 * nfull) + 4 * g;
        const bool ka = k0 < K, kb = full && k0 + 16 < K;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            xb[rt][0] = (rok[rt] && ka) ? xb[rt][0] : zero4;
            xb[rt][1] = (rok[rt] && kb) ? xb[rt][1] : zero4;
        }
    };
    f32x4 xa[RT][2], xb[RT][2], xc[RT][2];
    const int tile_stride = gridDim.x * kLxNW;
    int tile = blockIdx.x * kLxNW + wave;
    set_tile(tile < ntiles ? tile : 0);
    load_raw(0, xa);
    load_raw(1, xb);
    for (; tile < ntiles; tile += tile_stride) {
        f32x4 acc[MT][RT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(Bs + 16 * mt + 4 * g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
        }
        auto step = [&](int S, f32x4 (&cur)[RT][2], f32x4 (&nxt)[RT][2], f32x4 (&nn)[RT][2]) __attribute__((always_inline)) {
            load_raw(S + 2, nn);
            __builtin_amdgcn_sched_barrier(0);                // the loads stay HERE (the scheduler sinks them towards their use)
            lx_kstep<MT>(afrag + S * SL, cur, acc);
            finish_x(S + 1, nxt);                             // S + 1 past the end: finishes values nobody reads
        };
        finish_x(0, xa);
        int S = 0;
        for (; S + 3 <= nfull; S += 3) {
            step(S, xa, xb, xc);
            step(S + 1, xb, xc, xa);
            step(S + 2, xc, xa, xb);
        }
        if (nfull - S == 0) {
            if (tail) lx_ktail<MT>(afrag_t, xa, acc);
        } else if (nfull - S == 1) {
            step(S, xa, xb, xc);
            if (tail) lx_ktail<MT>(afrag_t, xb, acc);
        } else {
            step(S, xa, xb, xc);
            step(S + 1, xb, xc, xa);
            if (tail) lx_ktail<MT>(afrag_t, xc, acc);
        }
        int row_e[RT];
        bool rok_e[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { row_e[rt] = row[rt]; rok_e[rt] = rok[rt]; }
        set_tile(tile + tile_stride < ntiles ? tile + tile_stride : tile);
        load_raw(0, xa);
        load_raw(1, xb);
        __builtin_amdgcn_sched_barrier(0);
        lx_store<MT, GATE>(epi, acc, row_e, rok_e, g);
    }
}


// =====================================================================================================================================================
// The chunked form (any K, r6): the LDS image holds kLxKC k-steps of the weights at a time; per ROUND (one tile per wave) the workgroup walks the chunks —
// barrier, re-split the chunk (36 weights per thread), barrier, multiply — with the accumulators of the tile in registers across the chunks.  Serves the layers
// whose whole-K image does not fit (512 -> 136, the backward-input of the Q|K|V projection over 408) and 256 -> 512 (8 output tiles x 4 blocks instead of 6 x 6).
// The X fragments of a chunk (kLxKC x 16 registers) are requested in front of the staging and consumed behind it.
constexpr int kLxKC = 4;
template <int MT, bool TRANS, bool GATE>
__global__ void __launch_bounds__(kLxNT)
linear_fwd_x6c_kernel(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ gate, LinArgs a,
                      float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_lx[];
    constexpr int RT = kLxRT, KC = kLxKC;
    constexpr int PL = MT * 1024, SL = 3 * PL;
    const int K = a.K, R = a.R;
    const int K16 = (K + 15) & ~15, nfull = K16 >> 5;
    const bool tail = (K16 & 31) != 0;
    const int nsteps = nfull + (tail ? 1 : 0), nchunks = (nsteps + KC - 1) / KC;
    const int n0 = blockIdx.y * 16 * MT;
    float *Bs = reinterpret_cast<float *>(smem_lx + (size_t)KC * SL);
    lx_stage_bias<MT>(bias, a.N, n0, Bs);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int ntiles = (R + 16 * RT - 1) / (16 * RT);
    const LxEpi epi = lx_epi(a, gate, Y, n0);
    const uint32_t afrag = lx_lds_addr(smem_lx) + g * 256 + j * 16;
    const uint32_t afrag_t = lx_lds_addr(smem_lx) + g * 128 + j * 8;               // + (local step) SL
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int tiles_per_round = gridDim.x * kLxNW;
    const int nrounds = (ntiles + tiles_per_round - 1) / tiles_per_round;          // every wave walks every round: the chunk loop holds workgroup barriers
    for (int round = 0; round < nrounds; ++round) {
        const int tile = round * tiles_per_round + blockIdx.x * kLxNW + wave;
        int row[RT];
        const float *xrow[RT];
        bool rok[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            row[rt] = tile * 16 * RT + 16 * rt + j;                                 // a wave without a tile: rows past R, nothing loaded for real, nothing stored
            rok[rt] = tile < ntiles && row[rt] < R;
            xrow[rt] = X + (size_t)(rok[rt] ? row[rt] : R - 1) * a.ldx;
        }
        f32x4 acc[MT][RT];
        for (int c = 0; c < nchunks; ++c) {
            // ---- this chunk's X fragments (clamped addresses, selected below), in front of the staging
            f32x4 xs[KC][RT][2];
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int S = KC * c + s;
                const bool full = S < nfull, last = tail && S == nfull;
                const int k0 = (full ? 32 * S : 32 * nfull) + 4 * g;
                const int ka = ((full || last) && k0 < K) ? k0 : 0, kb = (full && k0 + 16 < K) ? k0 + 16 : 0;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    xs[s][rt][0] = *reinterpret_cast<const f32x4 *>(xrow[rt] + ka);
                    xs[s][rt][1] = *reinterpret_cast<const f32x4 *>(xrow[rt] + kb);
                }
            }
            __syncthreads();                                                        // every wave is done with the previous chunk's image
            const int steps_here = min(KC, nsteps - KC * c), full_here = min(max(nfull - KC * c, 0), KC);
            lx_stage<MT, TRANS>(W, K, a.N, n0, full_here, steps_here, smem_lx, 8 * KC * c);
            __syncthreads();
            if (c == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(Bs + 16 * mt + 4 * g);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
                }
            }
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int S = KC * c + s;
                if (S >= nsteps) break;                                             // uniform
                const bool full = S < nfull;
                const int k0 = (full ? 32 * S : 32 * nfull) + 4 * g;
                f32x4 x[RT][2];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    x[rt][0] = (rok[rt] && k0 < K) ? xs[s][rt][0] : zero4;
                    x[rt][1] = (rok[rt] && full && k0 + 16 < K) ? xs[s][rt][1] : zero4;
                }
                if (full) lx_kstep<MT>(afrag + s * SL, x, acc);
                else lx_ktail<MT>(afrag_t + s * SL, x, acc);
            }
        }
        lx_store<MT, GATE>(epi, acc, row, rok, g);
    }
}

}  // namespace

int launch_linear_x6(bool trans, const float *X, const float *W, const float *bias, const float *gate, const LinArgs &a, float *Y, int num_cus,
                     hipStream_t st, const char *who) {
    const char *sw = getenv("PTR_LIN_X6");                     // 0: fp32-MFMA kernel; 2: general form only (A/B measurements, tests; read per call)
    const int on = sw ? atoi(sw) : 1;
    if (!on) return -1;
    const int K = a.K, N = a.N, R = a.R;
    // served: K <= 256 in 16-byte rows (the fp32 X fragments are 16-byte loads), weights readable as 16-byte rows in the forward orientation
    if ((K & 3) || (a.ldx & 3) || (reinterpret_cast<uintptr_t>(X) & 15)) return -1;
    if (!trans && (reinterpret_cast<uintptr_t>(W) & 15)) return -1;
    if (R < 1024) return -1;                                  // small batches: the fp32 kernel's 16-row tiles fill the machine better
    // outputs (and the gate) as 16-byte pieces through 32-bit buffer offsets: aligned, N % 4 == 0, below 2 GB
    if ((N & 3) || (a.ldy & 3) || (reinterpret_cast<uintptr_t>(Y) & 15) || (size_t)R * a.ldy * 4 >= 0x80000000ull) return -1;
    const bool gated = a.act == PTR_LINEAR_GATE;
    if (gated && (!trans || (a.ldg & 3) || (reinterpret_cast<uintptr_t>(gate) & 15) || (size_t)R * a.ldg * 4 >= 0x80000000ull)) return -1;
    const int n16 = (N + 15) / 16;
    if (n16 < 4) return -1;                                   // narrow outputs stay with the fp32 kernel
    int per_tile = lx_steps(K) * 3 * 1024;                    // LDS bytes of one 16-row tile of the whole-K weight image
    int mt_max = (150 * 1024) / (per_tile + 64);
    if (mt_max > 9) mt_max = 9;
    int nby = mt_max >= 4 ? (n16 + mt_max - 1) / mt_max : 99;
    bool chunked = false;
    // X streams once per block of outputs: up to 3 blocks with the whole-K image, 4 when it still holds >= 8 tiles (136 -> 512: chunking a K of 4.5 k-steps
    // re-splits the weights twice per round for half a k-step of work — measured slower than the fp32 kernel); otherwise chunk K
    if (mt_max < 4 || nby > (mt_max >= 8 ? 4 : 3)) {
        chunked = true;
        per_tile = kLxKC * 3 * 1024;
        mt_max = n16 <= 9 ? 9 : 8;                            // 9 x 12 KB = 108 KB; wider outputs in blocks of 8 tiles
        nby = (n16 + mt_max - 1) / mt_max;
        if (nby > 4 || K <= 64) return -1;
    }
    const int MT = (n16 + nby - 1) / nby;
    if (MT < 4) return -1;
    const size_t lds = (size_t)MT * per_tile + (size_t)64 * MT;
    const int ntiles = (R + 16 * kLxRT - 1) / (16 * kLxRT);
    int gx = num_cus / nby;
    if (gx < 1) gx = 1;
    if (ntiles < gx * kLxNW) gx = (ntiles + kLxNW - 1) / kLxNW;
    auto go = [&](auto kern) -> int {
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(gx, nby), dim3(kLxNT), lds, st, X, W, bias, gate, a, Y);
        return check_hip(hipGetLastError(), who);
    };
    const int K16 = (K + 15) & ~15, nfull = K16 >> 5, tail = (K16 & 31) != 0;
    if (chunked) {
#define LXC_CASE(M)                                                                                                                             \
    case M:                                                                                                                                      \
        return !trans ? go(linear_fwd_x6c_kernel<M, false, false>) : (gated ? go(linear_fwd_x6c_kernel<M, true, true>) : go(linear_fwd_x6c_kernel<M, true, false>));
        switch (MT) { LXC_CASE(4) LXC_CASE(5) LXC_CASE(6) LXC_CASE(7) LXC_CASE(8) LXC_CASE(9) }
#undef LXC_CASE
        return -1;
    }
    // instantiated: forward (no gate), backward-input without and with the gate
    if (on != 2 && MT >= 7) {                                 // the whole-tile form: 97..144 inputs (K16 = 112, 128, 144), 7..9 output tiles
#define LXT_CASE(M, NF, TL)                                                                                                                     \
    if (MT == M && nfull == NF && tail == TL)                                                                                                    \
        return !trans ? go(linear_fwd_x6t_kernel<M, false, false, NF, TL != 0>)                                                                  \
                      : (gated ? go(linear_fwd_x6t_kernel<M, true, true, NF, TL != 0>) : go(linear_fwd_x6t_kernel<M, true, false, NF, TL != 0>));
        LXT_CASE(7, 3, 1) LXT_CASE(8, 3, 1) LXT_CASE(9, 3, 1)
        LXT_CASE(7, 4, 0) LXT_CASE(8, 4, 0) LXT_CASE(9, 4, 0)
        LXT_CASE(7, 4, 1) LXT_CASE(8, 4, 1) LXT_CASE(9, 4, 1)
#undef LXT_CASE
    }
#define LX_CASE(M)                                                                                                                              \
    case M:                                                                                                                                      \
        return !trans ? go(linear_fwd_x6_kernel<M, false, false>) : (gated ? go(linear_fwd_x6_kernel<M, true, true>) : go(linear_fwd_x6_kernel<M, true, false>));
    switch (MT) { LX_CASE(4) LX_CASE(5) LX_CASE(6) LX_CASE(7) LX_CASE(8) LX_CASE(9) }
#undef LX_CASE
    return -1;
}

}  // namespace ptr
