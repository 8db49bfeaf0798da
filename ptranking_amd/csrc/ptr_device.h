// Device + host helpers shared by every kernel of the ltr_adhoc hot path (gfx950 / CDNA4 only).
//
// Execution model used throughout: one "group" of G threads (G = 64: one wavefront, or G = 256: four) owns one
// query; a 256-thread workgroup therefore holds 256/G queries.  Per-query tiles (scores, gains, discounts, partial
// gradients) live in LDS; each thread owns the documents i = t, t+G, t+2G, ... (DPT of them, compile-time bound).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptranking_amd.h"

namespace ptr {

constexpr int kWave = 64;
constexpr int kBlock = 256;

// ---------------------------------------------------------------- host side: errors + launch plumbing
void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Shapes: L in (0, PTR_MAX_LIST_LEN]; B >= 0.  Returns 0 or an error code (message set).
int check_batch(const void *preds, const void *second, int B, int L, const char *who);

// (G, DPT) tiling by list length: G*DPT >= round_up(L, 4) always holds.
struct Tiling { int G; int DPT; };
inline Tiling pick_tiling(int L) {
    if (L <= 64) return {64, 1};
    if (L <= 128) return {64, 2};
    if (L <= 256) return {256, 1};
    if (L <= 512) return {256, 2};
    if (L <= 1024) return {256, 4};
    if (L <= 2048) return {256, 8};
    return {256, 16};
}

// Calls f.template operator()<G, DPT>() for the tiling of L.
template <class F> inline int dispatch_tiling(int L, F &&f) {
    Tiling t = pick_tiling(L);
    if (t.G == 64 && t.DPT == 1) return f.template operator()<64, 1>();
    if (t.G == 64 && t.DPT == 2) return f.template operator()<64, 2>();
    if (t.DPT == 1) return f.template operator()<256, 1>();
    if (t.DPT == 2) return f.template operator()<256, 2>();
    if (t.DPT == 4) return f.template operator()<256, 4>();
    if (t.DPT == 8) return f.template operator()<256, 8>();
    return f.template operator()<256, 16>();
}

// Tiling of the sort / metric kernels: ONE wavefront per query up to 1024 documents (the rank step is a register bitonic sort of
// 64*DPT keys, ptr_device.h wave_sort_desc), the four-wave counting form beyond.
template <class F> inline int dispatch_wave_tiling(int L, F &&f) {
    if (L <= 64) return f.template operator()<64, 1>();
    if (L <= 128) return f.template operator()<64, 2>();
    if (L <= 256) return f.template operator()<64, 4>();
    if (L <= 512) return f.template operator()<64, 8>();
    if (L <= 1024) return f.template operator()<64, 16>();
    if (L <= 2048) return f.template operator()<256, 8>();
    return f.template operator()<256, 16>();
}

template <class F> inline int dispatch_wave256_tiling(int L, F &&f) {
    if (L <= 64) return f.template operator()<64, 1>();
    if (L <= 128) return f.template operator()<64, 2>();
    if (L <= 256) return f.template operator()<64, 4>();
    return dispatch_tiling(L, f);
}

// Grid of a PERSISTENT kernel (its wavefronts walk the work with a grid stride): the blocks the device keeps resident at once — CUs x the
// occupancy of `kernel` at `block` threads and `lds` bytes of dynamic LDS — capped by the blocks the work needs.  Cached per kernel.
template <class K> inline int persistent_grid(K kernel, int block, size_t lds, int want) {
    // Keyed by (kernel, device, block, lds) and thread_local: kernels of one signature share this instantiation, a process may drive several
    // GPUs (data parallel) and several host threads (unsynchronised statics raced / reused another device's occupancy)
    struct Entry { const void *kernel; int dev, block; size_t lds; int resident; };
    thread_local Entry cache[8] = {};
    thread_local int next_slot = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const void *kp = reinterpret_cast<const void *>(kernel);
    int resident = 0;
    for (const Entry &e : cache)
        if (e.kernel == kp && e.dev == dev && e.block == block && e.lds == lds && e.resident > 0) { resident = e.resident; break; }
    if (resident <= 0) {
        int per_cu = 0;
        hipDeviceProp_t pr;
        int cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kp, block, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
        resident = cus * per_cu;
        cache[next_slot] = Entry{kp, dev, block, lds, resident};
        next_slot = (next_slot + 1) % 8;
    }
    static const int mult = [] { const char *e = getenv("PTR_PERSIST_MULT"); return e ? atoi(e) : 1; }();   // 0: one block per unit of work (measurements)
    if (mult <= 0) return want > 0 ? want : 1;
    const long cap = (long)resident * mult;
    return want < cap ? (want > 0 ? want : 1) : (int)cap;
}

// Raises the dynamic-LDS cap of `kernel` when a launch needs more than the 64 KiB default.
template <class K> inline int allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "hipFuncSetAttribute");
}

// ---------------------------------------------------------------- device side
#if defined(__HIPCC__)

__device__ __forceinline__ int query_len(const int32_t *lens, int q, int L) {
    if (!lens) return L;
    int n = lens[q];
    return n < 0 ? 0 : (n > L ? L : n);
}

// Butterfly reductions over one wavefront: every lane ends with the bit-identical result.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// Sum over the G threads of one group, fixed order (deterministic).  For G == 256 the group IS the workgroup, so the
// barriers inside are workgroup barriers: every thread of the block must call it.  `red` = 4 floats of LDS.
template <int G> __device__ __forceinline__ float group_sum(float v, float *red, int t) {
    if constexpr (G == 32) {               // two groups per wavefront: butterfly inside the 32-lane half, every lane of the half gets the sum
        (void)red; (void)t;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    }
    v = wave_sum(v);
    if constexpr (G == kWave) {
        return v;
    } else {
        __syncthreads();
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        float r = red[0];
#pragma unroll
        for (int w = 1; w < G / kWave; ++w) r += red[w];
        return r;
    }
}
// Deterministic sum of x[0..n) by ONE workgroup of 1024 threads (the loss-slot sum of every *_fwd_bwd entry point and of the fused
// backward step: both call THIS function, so the fused step returns the bits of the separate call).  Fixed order: thread t owns the
// elements i = t (mod 1024); sixteen independent partial sums (i / 1024 mod 16) keep sixteen loads in flight — the round-1 form, 256
// threads each walking a dependent load-add chain, took 60 us for 65 536 slots, longer than the ListNet kernel whose slots it sums (VERDICT
// r3) — combined as a balanced tree, then a wave butterfly, then the sixteen wave totals in order.  `red` = 16 floats of LDS; every
// thread of the block must call it (one barrier inside); the result is valid in thread 0.
__device__ __forceinline__ float block1024_sum(const float *__restrict__ x, int n, float *red) {
    const int t = threadIdx.x;
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.0f;
    int i = t;
    for (; i + 15 * 1024 < n; i += 16 * 1024) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] += x[i + u * 1024];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { if (i + u * 1024 < n) a[u] += x[i + u * 1024]; }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) a[u] += a[u + w];
    const float v = wave_sum(a[0]);
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r += red[w];
    return r;
}
template <int G> __device__ __forceinline__ float group_max(float v, float *red, int t) {
    v = wave_max(v);
    if constexpr (G == kWave) {
        return v;
    } else {
        __syncthreads();
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        float r = red[0];
#pragma unroll
        for (int w = 1; w < G / kWave; ++w) r = fmaxf(r, red[w]);
        return r;
    }
}

// Inclusive prefix scans over one wavefront (lane order) out of the VALU alone: DPP row shifts inside the 16-lane rows, then the row
// totals carried across rows by row_bcast:15 / :31 — 6 instructions per scan (the __shfl_up form goes through the LDS crossbar six
// times and needs a select per step).  Lanes without a source (row_shr beyond the row start, unwritten rows) take the identity.
// r5: written as v_add_f32 / v_mul_f32 with a DPP source — one instruction per step (the builtin form cost an identity move, the DPP move
// and the operation for the row_bcast steps and for every product step); a lane a step does not write (no source lane inside the row,
// rows outside row_mask) keeps its value, which is the identity's effect.  s_nop: the VALU-write -> DPP-read hazard (2 wait states) that
// the assembler does not see inside an asm block.  The *2 forms run two independent scans interleaved (one wait state filled by the
// other scan's instruction).  Same operations in the same order as before: bit-identical results.
#define PTR_SCAN_STEPS(op, n0)                                                          \
    n0 op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
    "s_nop 1"
#define PTR_SCAN2_STEP(op, ctl) op " %0, %0, %0 " ctl "\n\t" op " %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
#define PTR_SCAN2_STEPS(op)                                                             \
    "s_nop 1\n\t"                                                                      \
    PTR_SCAN2_STEP(op, "row_shr:1 row_mask:0xf bank_mask:0xf")                          \
    PTR_SCAN2_STEP(op, "row_shr:2 row_mask:0xf bank_mask:0xf")                          \
    PTR_SCAN2_STEP(op, "row_shr:4 row_mask:0xf bank_mask:0xf")                          \
    PTR_SCAN2_STEP(op, "row_shr:8 row_mask:0xf bank_mask:0xf")                          \
    PTR_SCAN2_STEP(op, "row_bcast:15 row_mask:0xa bank_mask:0xf")                       \
    PTR_SCAN2_STEP(op, "row_bcast:31 row_mask:0xc bank_mask:0xf")                       \
    "s_nop 0"
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
    (void)lane;
    asm(PTR_SCAN_STEPS("v_add_f32_dpp", "s_nop 1\n\t") : "+v"(v));
    return v;
}
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
    (void)lane;
    asm(PTR_SCAN_STEPS("v_mul_f32_dpp", "s_nop 1\n\t") : "+v"(v));
    return v;
}
__device__ __forceinline__ void wave_incl_sum2(float &a, float &b) { asm(PTR_SCAN2_STEPS("v_add_f32_dpp") : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void wave_incl_prod2(float &a, float &b) { asm(PTR_SCAN2_STEPS("v_mul_f32_dpp") : "+v"(a), "+v"(b)); }
#undef PTR_SCAN_STEPS
#undef PTR_SCAN2_STEP
#undef PTR_SCAN2_STEPS
// Inclusive SUFFIX sum over one wavefront (lane i gets sum of lanes i..63).
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v += o;
    }
    return v;
}

// ln(x) = log2(x)*ln2 on the transcendental pipe (v_log_f32); for x == 0 or normal x only (no denormal scaling).
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// 2^l - 1 (ptranking/metric/adhoc/adhoc_metric.py:208-209); exact for the integer grades 0..4 stored as floats.
__device__ __forceinline__ float gain_of(float label) { return exp2f(label) - 1.0f; }

// Descending rank of each owned key among keys[0..n): rank = #{j : k_j > k_i  or (k_j == k_i and j < i)}, i.e. the
// position torch.sort(descending=True) gives on tie-free input, with ties broken by original index.
// keys[] is in LDS, padded with -inf up to a multiple of 4 (float4 broadcast reads).  own[m] / index t + m*G.
// BLK: own[m] is document t*DPT + m (the blocked layout of the one-wavefront paths) instead of t + m*G.
template <int G, int DPT, bool BLK = false>
__device__ __forceinline__ void count_ranks(const float *keys, int n, int t, const float (&own)[DPT], int (&rk)[DPT]) {
    // Fast path (tie-free lists, the common case): rank = #{j : k_j > k_i} — one compare + one add-with-carry per pair.
    // #{j : k_j >= k_i} is counted alongside; a lane whose two counts differ by more than its own element has a tie, and
    // only then the wave takes the slow pass that adds #{j < i : k_j == k_i} (original index breaks ties).
    int ge[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) { rk[m] = 0; ge[m] = 0; }
    const float4 *k4 = reinterpret_cast<const float4 *>(keys);
    const int n4 = (n + 3) >> 2;
    for (int j4 = 0; j4 < n4; ++j4) {
        const float4 v = k4[j4];
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const float s = own[m];
            rk[m] += (v.x > s) + (v.y > s) + (v.z > s) + (v.w > s);
            ge[m] += (v.x >= s) + (v.y >= s) + (v.z >= s) + (v.w >= s);
        }
    }
    bool tie = false;
#pragma unroll
    for (int m = 0; m < DPT; ++m) tie |= ((BLK ? t * DPT + m : t + m * G) < n) && (ge[m] - rk[m] != 1);
    if (__any(tie)) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = BLK ? t * DPT + m : t + m * G;
            const float s = own[m];
            if (i < n && ge[m] - rk[m] != 1) {
                int extra = 0;
                for (int j = 0; j < i; ++j) extra += keys[j] == s ? 1 : 0;
                rk[m] += extra;
            }
        }
    }
}

// ---- packed rank counting (scores: tie-free in the common case)
using f32x2 = __attribute__((ext_vector_type(2))) float;
// v_pk_fma_f32 with the clamp modifier: {clamp(a.x*b.x+c.x, 0, 1), clamp(a.y*b.y+c.y, 0, 1)}
__device__ __forceinline__ f32x2 pk_fma_clamp(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// count_ranks() at ONE VALU slot per compare instead of four: t = clamp(BIG*k_j - BIG*k_i, 0, 1) is exactly 1 for k_j > k_i and 0 otherwise
// (the fma is exact up to its final rounding, so the sign is right and 0 means equal; two compares per v_pk_fma_f32), summed in fp32
// (exact up to 2^24).  BIG = 2^100: t is fractional only for 0 < k_j - k_i < 2^-100, and BIG*k overflows only for |k| >= 2^28 (inf - inf
// = NaN clamps to 0) — then the sums are not all integers, or two documents share a rank, as they do for TIES (equal keys: rank =
// original index order).  Both are detected (integrality + a scatter / gather permutation check through `mark`, n ints of LDS), and
// the whole group recounts with count_ranks().  keys[] as for count_ranks (padded with -inf to a multiple of 4).  Contains group-wide
// barriers: every thread of the group (G == 256: of the block) must call it.
template <int G, int DPT>
__device__ __forceinline__ void count_ranks_fast(const float *keys, int *mark, int n, int t, const float (&own)[DPT], int (&rk)[DPT]) {
    const float big = 0x1p100f;
    const f32x2 big2 = {big, big};
    f32x2 nsb[DPT], cnt[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const float v = t + m * G < n ? -own[m] * big : 0.0f;
        nsb[m] = f32x2{v, v}; cnt[m] = f32x2{0.0f, 0.0f};
    }
    const float4 *k4 = reinterpret_cast<const float4 *>(keys);
    // n is the same for every thread of the group: a scalar trip count (no exec-mask loop), 16 keys per trip with their LDS reads
    // issued ahead of the arithmetic, two accumulator chains per document
    const int n4 = __builtin_amdgcn_readfirstlane((n + 3) >> 2);
    f32x2 cnt2[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) cnt2[m] = f32x2{0.0f, 0.0f};
    auto body = [&](const float4 v) {
        const f32x2 u0 = {v.x, v.y}, u1 = {v.z, v.w};
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            cnt[m] += pk_fma_clamp(u0, big2, nsb[m]);
            cnt2[m] += pk_fma_clamp(u1, big2, nsb[m]);
        }
    };
    int j4 = 0;
    for (; j4 + 4 <= n4; j4 += 4) {
        const float4 v0 = k4[j4], v1 = k4[j4 + 1], v2 = k4[j4 + 2], v3 = k4[j4 + 3];
        body(v0); body(v1); body(v2); body(v3);
    }
    for (; j4 < n4; ++j4) body(k4[j4]);
    bool redo = false;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const float c = (cnt[m].x + cnt[m].y) + (cnt2[m].x + cnt2[m].y);
        rk[m] = (int)c;
        redo |= i < n && ((float)rk[m] != c || rk[m] >= n || rk[m] < 0);
        if (i < n && rk[m] >= 0 && rk[m] < n) mark[rk[m]] = i;
    }
    if constexpr (G == kWave) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        redo |= i < n && rk[m] >= 0 && rk[m] < n && mark[rk[m]] != i;
    }
    bool any;
    if constexpr (G == kWave) any = __any(redo);
    else any = __syncthreads_or(redo);
    if (any) count_ranks<G, DPT>(keys, n, t, own, rk);
}

// ---- helpers of the register "ring" pair loops (pairwise.hip lambdarank_ring_kernel, approxndcg.hip approxndcg_ring_kernel)
// a - b as ONE packed instruction (the compiler splits a v2f32 subtraction whose lanes are consumed separately)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// Wavefront sum out of the VALU alone (DPP row operations + one v_readlane; the butterfly wave_sum() goes through the LDS
// crossbar six times, latency the four co-resident waves cannot hide because they run the same phase).  Fixed order; every lane
// receives the same value.
#define PTR_DPP_ADD(v, ctrl, rows) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
    PTR_DPP_ADD(v, 0xB1, 0xF);        // quad_perm [1,0,3,2]
    PTR_DPP_ADD(v, 0x4E, 0xF);        // quad_perm [2,3,0,1]
    PTR_DPP_ADD(v, 0x141, 0xF);       // row_half_mirror
    PTR_DPP_ADD(v, 0x140, 0xF);       // row_mirror: every lane holds its row's sum
    PTR_DPP_ADD(v, 0x142, 0xA);       // row_bcast:15 into rows 1 and 3
    PTR_DPP_ADD(v, 0x143, 0xC);       // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef PTR_DPP_ADD
// Wavefront max / integer min out of the VALU alone, same DPP ladder (lanes a control does not write keep the identity); every lane receives
// the result (v_readlane of lane 63).
#define PTR_DPP_STEP(v, ident, ctrl, rows) __builtin_amdgcn_update_dpp((ident), (v), (ctrl), (rows), 0xF, false)
__device__ __forceinline__ float wave_max_dpp(float x) {
    // one v_max_f32 with a DPP source per step (r5: the builtin form cost five instructions a step — the identity for the lanes without a
    // source, the move and two canonicalising maxima); lanes a step does not write (row_mask, no source lane) keep their value.  The
    // s_nop covers the VALU-write -> DPP-read hazard the assembler does not see inside an asm block.
    float v = x;
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
    const int big = 0x7fffffff;
    v = min(v, PTR_DPP_STEP(v, big, 0xB1, 0xF));
    v = min(v, PTR_DPP_STEP(v, big, 0x4E, 0xF));
    v = min(v, PTR_DPP_STEP(v, big, 0x141, 0xF));
    v = min(v, PTR_DPP_STEP(v, big, 0x140, 0xF));
    v = min(v, PTR_DPP_STEP(v, big, 0x142, 0xA));
    v = min(v, PTR_DPP_STEP(v, big, 0x143, 0xC));
    return __builtin_amdgcn_readlane(v, 63);
}
#undef PTR_DPP_STEP
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// 1 / log2(pos + 2)
__device__ __forceinline__ float inv_log2_pos(int pos) {
    const float d = __builtin_amdgcn_logf((float)(pos + 2));
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.0f), r);
}
// LDS hand-over between the lanes of ONE wavefront (its own LDS region): LDS operations of a wave execute in order, only the
// compiler has to be kept from reordering them — no workgroup barrier, the four waves of a block stay independent
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#ifndef PTR_RING_DPP
#define PTR_RING_DPP 0x134
#endif
__device__ __forceinline__ float dpp_rol1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), PTR_RING_DPP /* wave_rol:1 */, 0xF, 0xF, false));
}


// ---- one wavefront sorts 64*E keys held E per lane (bitonic network in registers)
// Value of lane (lane ^ X): DPP inside the 16-lane rows where a control exists (quad_perm, row_half_mirror, row_mirror, row_ror:8),
// the LDS crossbar otherwise (ds_swizzle bit mode inside 32 lanes, ds_bpermute across the halves).
template <int X> __device__ __forceinline__ int lane_xor_bits(int s, int lane) {
    if constexpr (X == 1) return __builtin_amdgcn_mov_dpp(s, 0xB1, 0xF, 0xF, true);            // quad_perm:[1,0,3,2]
    else if constexpr (X == 2) return __builtin_amdgcn_mov_dpp(s, 0x4E, 0xF, 0xF, true);       // quad_perm:[2,3,0,1]
    else if constexpr (X == 3) return __builtin_amdgcn_mov_dpp(s, 0x1B, 0xF, 0xF, true);       // quad_perm:[3,2,1,0]
    else if constexpr (X == 7) return __builtin_amdgcn_mov_dpp(s, 0x141, 0xF, 0xF, true);      // row_half_mirror
    else if constexpr (X == 15) return __builtin_amdgcn_mov_dpp(s, 0x140, 0xF, 0xF, true);     // row_mirror
    else if constexpr (X == 8) return __builtin_amdgcn_mov_dpp(s, 0x128, 0xF, 0xF, true);      // row_ror:8
    else if constexpr (X == 4 || X == 16 || X == 31) return __builtin_amdgcn_ds_swizzle(s, (X << 10) | 0x1F);   // bit mode: and 0x1F, xor X
    else return __builtin_amdgcn_ds_bpermute((lane ^ X) << 2, s);
}
// (every source lane of these controls is inside the row: v_mov_b32_dpp without an `old` operand — the tied form cost a register copy
// per exchange, r5)
template <int X, class T> __device__ __forceinline__ T lane_xor(T v, int lane) {
    return __builtin_bit_cast(T, lane_xor_bits<X>(__builtin_bit_cast(int, v), lane));
}
constexpr int top_bit(int x) { int b = 1; while (b * 2 <= x) b *= 2; return b; }
// The two key types of the network: float (v_med3_f32 against +-inf is max / min without the canonicalising extra instruction
// fmaxf / fminf bring, and returns one of its inputs bit for bit; NaNs are screened before the sort) and uint32_t (packed integer keys).
__device__ __forceinline__ float sort_hi(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
__device__ __forceinline__ float sort_lo(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }
__device__ __forceinline__ uint32_t sort_hi(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t sort_lo(uint32_t a, uint32_t b) { return a < b ? a : b; }
// bound: the key type's maximum where the lane keeps the larger key, its minimum where it keeps the smaller one
__device__ __forceinline__ float sort_bound(float, bool keep_min) { return __builtin_bit_cast(float, 0x7F800000 | (keep_min ? (int)0x80000000 : 0)); }
__device__ __forceinline__ uint32_t sort_bound(uint32_t, bool keep_min) { return keep_min ? 0u : 0xFFFFFFFFu; }
__device__ __forceinline__ float sort_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ uint32_t sort_med3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// Compare-exchange of positions p and p ^ X inside the lane's registers (X < E); the lower position keeps the larger key.
template <int E, int X, class T> __device__ __forceinline__ void bitonic_local(T (&v)[E]) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if ((r & top_bit(X)) == 0) {
            const T a = v[r], b = v[r ^ X];
            v[r] = sort_hi(a, b);
            v[r ^ X] = sort_lo(a, b);
        }
    }
}
// Compare-exchange with lane (lane ^ LX); FLIP: against the partner's register E-1-r (the mirror step of a merge), else register r.
// The lane whose top differing bit is clear keeps the maximum: one med3 against a per-lane bound.
template <int E, int LX, bool FLIP, class T> __device__ __forceinline__ void bitonic_cross(T (&v)[E], int lane) {
    const T bound = sort_bound(T(), (lane & top_bit(LX)) != 0);
    T p[E];
#pragma unroll
    for (int r = 0; r < E; ++r) p[r] = lane_xor<LX>(v[FLIP ? E - 1 - r : r], lane);
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = sort_med3(v[r], p[r], bound);
}
template <int E, int J, class T> __device__ __forceinline__ void bitonic_halves(T (&v)[E], int lane) {   // half-cleaners J, J/2, ..., 1
    if constexpr (J >= 1) {
        if constexpr (J < E) bitonic_local<E, J>(v);
        else bitonic_cross<E, J / E, false>(v, lane);
        bitonic_halves<E, J / 2>(v, lane);
    }
}
template <int E, int K, class T> __device__ __forceinline__ void bitonic_phases(T (&v)[E], int lane) {   // merges of size K, 2K, ..., 64E
    if constexpr (K <= 64 * E) {
        if constexpr (K <= E) bitonic_local<E, K - 1>(v);                   // mirror step p <-> p ^ (K-1): every direction is "descending"
        else bitonic_cross<E, K / E - 1, true>(v, lane);
        bitonic_halves<E, K / 4>(v, lane);
        bitonic_phases<E, K * 2>(v, lane);
    }
}
// Sorts the 64*E keys v[r] = key at position lane*E + r into descending order (position 0 = maximum).  float keys: no NaNs.  36
// compare-exchange stages for 256 keys = ~230 VALU / crossbar instructions per wavefront, against 65 536 / 64 packed compares of the
// counting form.
template <int E, class T> __device__ __forceinline__ void wave_sort_desc(T (&v)[E], int lane) { bitonic_phases<E, 2>(v, lane); }

// count_ranks_fast() for a group of ONE wavefront: sort the keys, then each document finds its rank in the sorted row by binary search
// (rank = #{k_j > k_i}); ties among the n valid keys (adjacent equal entries) and NaNs send the wave to the exact count_ranks().
// keys[]: LDS, entries [0, Lp) (Lp = round_up(n.., 4), padded with -inf beyond n); sorted[]: LDS scratch of 64*DPT floats, left holding
// the keys in descending order (-inf beyond n).  own[m] / index t + m*64 as everywhere.  Wave-level barriers only.
template <int DPT>
__device__ __forceinline__ void count_ranks_wave(const float *keys, float *sorted, int n, int Lp, int t, const float (&own)[DPT], int (&rk)[DPT]) {
    constexpr int N = kWave * DPT;
    float v[DPT];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < DPT; ++r) {
        const int p = t * DPT + r;
        v[r] = p < Lp ? keys[p] : -INFINITY;
        bad |= v[r] != v[r];
    }
    wave_sort_desc<DPT>(v, t);
    const float nxt = __shfl_down(v[0], 1, 64);                              // first key of the next lane (lane 63: unused, its pairs end past n)
#pragma unroll
    for (int r = 0; r < DPT; ++r) {
        const int p = t * DPT + r;
        sorted[p] = v[r];
        bad |= p + 1 < n && v[r] == (r + 1 < DPT ? v[(r + 1) % DPT] : nxt);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (__any(bad)) { count_ranks<kWave, DPT>(keys, n, t, own, rk); return; }
#pragma unroll
    for (int m = 0; m < DPT; ++m) rk[m] = 0;
#pragma unroll
    for (int step = N / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) rk[m] += sorted[rk[m] + step - 1] > own[m] ? step : 0;
    }
}

// r5: count_ranks_wave() for documents held in the BLOCKED layout — lane t owns documents t*DPT .. t*DPT + DPT-1 (16-byte global loads,
// no LDS staging of the keys before the sort): own[r] = key of document t*DPT + r (-inf beyond n).  Leaves v[] = the keys in descending
// order (v[r] = position t*DPT + r; garbage if a key is NaN), sorted[] (LDS, 64*DPT floats) = the same row, rk[r] = rank of document
// t*DPT + r.  Ties / NaNs: the keys go to `scratch` (LDS, 64*DPT floats) and the wave recounts exactly.  Wave-level barriers only; both
// LDS rows may be overwritten by the caller on return.
template <int DPT> __device__ __forceinline__ void lds_store_blocked(float *row, int t, const float (&v)[DPT]) {
    if constexpr (DPT % 4 == 0) {
#pragma unroll
        for (int r = 0; r < DPT; r += 4) *reinterpret_cast<float4 *>(row + t * DPT + r) = float4{v[r], v[r + 1], v[r + 2], v[r + 3]};
    } else if constexpr (DPT == 2) {
        *reinterpret_cast<float2 *>(row + t * 2) = float2{v[0], v[1]};
    } else {
#pragma unroll
        for (int r = 0; r < DPT; ++r) row[t * DPT + r] = v[r];
    }
}
template <int DPT>
__device__ __forceinline__ void rank_blocked_wave(float *sorted, float *scratch, int n, int t, const float (&own)[DPT], int (&rk)[DPT],
                                                  float (&v)[DPT]) {
    constexpr int N = kWave * DPT;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < DPT; ++r) { v[r] = own[r]; bad |= v[r] != v[r]; }
    wave_sort_desc<DPT>(v, t);
    const float nxt = __shfl_down(v[0], 1, 64);                              // first key of the next lane (lane 63: unused, its pairs end past n)
#pragma unroll
    for (int r = 0; r < DPT; ++r) bad |= t * DPT + r + 1 < n && v[r] == (r + 1 < DPT ? v[(r + 1) % DPT] : nxt);
    lds_store_blocked<DPT>(sorted, t, v);
    if (__any(bad)) {
        lds_store_blocked<DPT>(scratch, t, own);
        wave_lds_sync();
        count_ranks<kWave, DPT, true>(scratch, n, t, own, rk);
        wave_lds_sync();
        return;
    }
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < DPT; ++m) rk[m] = 0;
#pragma unroll
    for (int step = N / 2; step >= 1; step >>= 1) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) rk[m] += sorted[rk[m] + step - 1] > own[m] ? step : 0;
    }
    wave_lds_sync();
}
__device__ __forceinline__ float dpp_wave_shl1(float v) {      // lane t <- lane t + 1, lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_shr1(float v) {      // lane t <- lane t - 1, lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ int dpp_wave_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }

// r5: the (score descending, index ascending) order of 65 .. 1024 documents from ONE register sort, without a rank search.
// Key = order-preserving integer image of the score with its low IB bits replaced by (N - 1 - index) (N = 64 DPT, IB = log2 N), sorted
// descending by the integer network; position p's document is read off the key, its true score gathered from the row staged in LDS.
// The truncated order is wrong only between documents whose scores agree in the top 32 - IB bits (about one pair in eight lists of 256
// N(0,1) scores, a handful of pairs in a list of 1024): one odd-even transposition round on the true (score, index) pairs repairs isolated pairs, every adjacent pair is then
// CHECKED — a sorted row is exactly one whose adjacent pairs are ordered — and the wave returns false (caller: the float sort + search /
// exact count) if any is not, or if a score is NaN.  -0.0 keys as +0.0 (torch.sort compares them equal; the index decides).
// own[r] = score of document t*DPT + r (-inf beyond n); raw: LDS, N floats (left holding the scores by document).  On success sc[r] / id[r]
// = score / document of position t*DPT + r (positions >= n: -inf / N-1).
template <int DPT, bool NEED_SC>
__device__ __forceinline__ bool sort_scores_packed(float *raw, int n, int t, const float (&own)[DPT], float (&sc)[DPT], int (&id)[DPT]) {
    static_assert(DPT == 2 || DPT == 4 || DPT == 8 || DPT == 16, "lists of 65 .. 1024 documents");
    constexpr int N = kWave * DPT;
    uint32_t key[DPT];
    bool bad = false;
    auto build = [&](auto full) {
#pragma unroll
        for (int r = 0; r < DPT; ++r) {
            const int i = t * DPT + r;
            const float x = own[r] + 0.0f;
            bad |= x != x;
            const int b = __builtin_bit_cast(int, x);
            const uint32_t o = (uint32_t)b ^ ((uint32_t)(b >> 31) | 0x80000000u);
            const uint32_t kv = (o & ~(uint32_t)(N - 1)) | (uint32_t)(N - 1 - i);
            key[r] = (decltype(full)::value || i < n) ? kv : 0u;
        }
    };
    if (n == N) build(std::true_type{}); else build(std::false_type{});     // uniform: full lists carry no padding masks
    lds_store_blocked<DPT>(raw, t, own);
    if (__any(bad)) return false;
    wave_sort_desc<DPT>(key, t);
    wave_lds_sync();
    // documents whose scores agree in the key's score bits sit next to each other: no such pair among the n documents -> the order is exact
    bool coll = false;
#pragma unroll
    for (int r = 0; r + 1 < DPT; ++r) coll |= (key[r] ^ key[r + 1]) < (uint32_t)N && t * DPT + r + 1 < n;
    const uint32_t nkey = (uint32_t)dpp_wave_shl1((int)key[0]);
    coll |= (key[DPT - 1] ^ nkey) < (uint32_t)N && t * DPT + DPT < n && t < 63;
#pragma unroll
    for (int r = 0; r < DPT; ++r) id[r] = N - 1 - (int)(key[r] & (uint32_t)(N - 1));
    const bool repair = __any(coll);
    if (NEED_SC || repair) {
#pragma unroll
        for (int r = 0; r < DPT; ++r) sc[r] = raw[id[r]];
    }
    if (!repair) return true;
    auto wrong = [](float sa, int ia, float sb, int ib) { return sa < sb || (sa == sb && ia > ib); };
    auto fix = [&](int a, int b) {
        const bool w = wrong(sc[a], id[a], sc[b], id[b]);
        const float s0 = w ? sc[b] : sc[a], s1 = w ? sc[a] : sc[b];
        const int i0 = w ? id[b] : id[a], i1 = w ? id[a] : id[b];
        sc[a] = s0; sc[b] = s1; id[a] = i0; id[b] = i1;
    };
#pragma unroll
    for (int r = 0; r + 1 < DPT; r += 2) fix(r, r + 1);                      // even pairs (p, p+1): inside the lane
#pragma unroll
    for (int r = 1; r + 1 < DPT; r += 2) fix(r, r + 1);                      // odd pairs inside the lane
    {                                                                        // the odd pair across the lane boundary
        const float ns = dpp_wave_shl1(sc[0]), ps = dpp_wave_shr1(sc[DPT - 1]);
        const int ni = dpp_wave_shl1(id[0]), pi = dpp_wave_shr1(id[DPT - 1]);
        const bool w_hi = t < 63 && wrong(sc[DPT - 1], id[DPT - 1], ns, ni), w_lo = t > 0 && wrong(ps, pi, sc[0], id[0]);
        if (w_hi) { sc[DPT - 1] = ns; id[DPT - 1] = ni; }
        if (w_lo) { sc[0] = ps; id[0] = pi; }
    }
    bool still = false;
#pragma unroll
    for (int r = 0; r + 1 < DPT; ++r) still |= wrong(sc[r], id[r], sc[r + 1], id[r + 1]);
    const float vs = dpp_wave_shl1(sc[0]);                                   // (outside the condition: a DPP read of a lane that a branch has
    const int vi = dpp_wave_shl1(id[0]);                                     //  switched off returns the bound value, not the lane's register)
    still |= t < 63 && wrong(sc[DPT - 1], id[DPT - 1], vs, vi);
    return !__any(still);
}

// 16-byte global loads of a row into the blocked layout (row base 16-byte aligned: L % 4 == 0 and `aligned` = the tensor's base is), scalar
// loads otherwise; `pad` beyond n
template <int DPT>
__device__ __forceinline__ void load_blocked(const float *__restrict__ row, int n, int L, int t, float pad, float (&x)[DPT], bool aligned = true) {
    if (DPT % 4 == 0 && n == kWave * DPT && aligned) {                                  // full row (n == L == 64 DPT): no masks
#pragma unroll
        for (int r = 0; r < DPT; r += 4) {
            const float4 u = *reinterpret_cast<const float4 *>(row + t * DPT + r);
            x[r] = u.x; x[r + 1] = u.y; x[r + 2] = u.z; x[r + 3] = u.w;
        }
    } else if (DPT % 4 == 0 && (L & 3) == 0 && aligned) {
#pragma unroll
        for (int r = 0; r < DPT; r += 4) {
            const int i = t * DPT + r;
            float4 u = float4{pad, pad, pad, pad};
            if (i < n) u = *reinterpret_cast<const float4 *>(row + i);
            x[r] = u.x; x[r + 1] = i + 1 < n ? u.y : pad; x[r + 2] = i + 2 < n ? u.z : pad; x[r + 3] = i + 3 < n ? u.w : pad;
        }
    } else {
#pragma unroll
        for (int r = 0; r < DPT; ++r) x[r] = t * DPT + r < n ? row[t * DPT + r] : pad;
    }
}

// Ideal-order staging shared by LambdaLoss / ApproxNDCG / the metric kernel.
// Every thread arrives with its own documents i = t + m*G (score si, label li; padded docs: -inf / 0) and leaves with
//   ipos[m]  = position of document i in the ideal (label-descending, index-ascending on ties) order,
//   S_id/Y_id = scores / labels by ideal position in LDS (Lp entries; padded tail: -inf / 0).
// presort != 0: the labels already are in ideal order (ipos = i) — the reference's `if presort:` branches
// (ptranking/ltr_adhoc/listwise/lambdaloss.py:83-87, approxNDCG.py:94-98, ptranking/base/ranker.py:53-56).
// Ends with a workgroup barrier; must be called by every thread of the block.
template <int G, int DPT>
__device__ __forceinline__ void stage_ideal_order(float *S_id, float *Y_id, int n, int Lp, int t, bool presort,
                                                  const float (&si)[DPT], const float (&li)[DPT], int (&ipos)[DPT]) {
    if (presort) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            ipos[m] = i;
            if (i < Lp) { S_id[i] = si[m]; Y_id[i] = li[m]; }
        }
        __syncthreads();
        return;
    }
    // keys = labels (padded with -inf) staged in S_id first, ranks counted, then both arrays rewritten in ideal order
    float ky[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        ky[m] = i < n ? li[m] : -INFINITY;
        if (i < Lp) S_id[i] = ky[m];
    }
    __syncthreads();
    count_ranks<G, DPT>(S_id, n, t, ky, ipos);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < n) { S_id[ipos[m]] = si[m]; Y_id[ipos[m]] = li[m]; }
        else if (i < Lp) { S_id[i] = -INFINITY; Y_id[i] = 0.0f; ipos[m] = i; }
        else ipos[m] = i;
    }
    __syncthreads();
}

#endif  // __HIPCC__
}  // namespace ptr
