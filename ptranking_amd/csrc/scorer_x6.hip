// Fused pointwise MLP scorer (the reference's `pointsf`) on bf16 matrix instructions with fp32 results ("bf16 x 6").
//
// Reference: ptranking/base/point_ranker.py:30-55, ptranking/base/utils.py:288-356 ((Dropout -> Linear -> ReLU) x NL -> Linear).
//
// Arithmetic.  An fp32 number is EXACTLY the sum of three bf16 pieces (8 + 8 + 8 mantissa bits; split by rounding to nearest, see
// split_pack2), so  a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + terms below 2^-24 |a b|:  six v_mfma_f32_16x16x32_bf16
// (fp32 accumulation inside the instruction) do the work of eight v_mfma_f32_16x16x4_f32 at 6 x 16 instead of 8 x 32 issue cycles, with
// an error against float64 equal to or below the fp32 MFMA's (scratch/bf16x6, tests/test_x6_gpu.py).  The bf16 matrix pipe also runs
// BESIDE the vector ALU (the fp32 MFMA shares its issue stream), so the dropout generator and the splitting ride in the MFMA gaps.
//
// Data flow ("transposed world", as scorer.hip):  Z^T[feature][doc] = W[feature][k] * A^T[k][doc].  A wave owns 32 documents (two
// 16-document B tiles) through all layers; an output tile leaves lane (j = l & 15, g = l >> 4) holding document j's features
// 16 mt + 4 g + {0..3}, which are split and packed IN REGISTERS into the next layer's B fragments: the contraction index of a 32-deep
// slice is visited in the order  slot (g, e) <-> feature 32 s + 16 (e >> 2) + 4 g + (e & 3)  (any order is fine as long as A and B agree),
// so activations never leave registers.  The weights are the A operands: pre-split into three bf16 planes by x6_prep_kernel (one small
// launch per call — the weights change every step) in exactly the image the fragments are read from, slice by slice:
//   slice = [3 planes][7 tiles][lane group g][16 out-features][8 k-slots] bf16 = 21 504 B;  layer 1: ceil(F / 32) slices (slot (g, e) <-> feature 32 s + 8 g + e,
//   what a lane loads from its X row as two float4), every further hidden layer 4 slices (100 -> 128, zero padded).
// All planes (F = 136, NL = 3: 13 slices = 280 KB) do not fit in the 160 KB LDS: the 8 waves of a workgroup walk the slices in LOCKSTEP
// while the image streams from L2 through a ring of kRing slices by LDS-DMA (global_load_lds_dwordx4, 3 KB per wave and slice), one
// barrier per slice; every slice is fetched once per workgroup and 256 documents.
#include "ptr_mlp.h"

namespace ptr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
union Frag { bf16x8 v; u32x4 q; uint32_t u[4]; };

// (kX6Rows / kX6PlaneBytes / kX6SliceBytes / x6_n1 / x6_nslices: ptr_mlp.h — the optimiser step of the fused train step writes the image too)
constexpr int kX6Pieces = kX6SliceBytes / 1024;
constexpr int kX6Ring = 6;                         // slices resident in LDS

__host__ __device__ inline size_t x6_lds_bytes(int NL) { return (size_t)kX6Ring * kX6SliceBytes + ((size_t)NL * kHP + kHP + 16) * sizeof(float); }

// Split by ROUNDING (v_cvt_pk_bf16_f32, round to nearest even): a = p1 + p2 + p3 exactly (|p2| <= 2^-9 |a|, |p3| <= 2^-18 |a|), and the
// pieces below the first carry either sign — the dropped products a2 b3 + a3 b2 + a3 b3 (<= 2^-26 |a b|) average out.  A split by
// truncation (two AND / SUB pairs, r3 probe) has all pieces of one sign: on all-positive data every dropped product pulls the same way
// and the median relative error was 1.6e-7 against 5e-8 for the fp32 MFMA (tests/test_x6_gpu.py all_positive); same instruction count.
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float x0, float x1) {                // {bf16(x0), bf16(x1)} in one dword, element 0 in the low half
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
// two fp32 values -> one dword of each of the three planes
__device__ __forceinline__ void split_pack2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk_bf16(s0, s1);
}
// four consecutive fp32 values -> dwords d, d+1 of the three plane fragments
__device__ __forceinline__ void split_pack4(const f32x4 v, Frag (&f)[3], int d) {
    split_pack2(v[0], v[1], f[0].u[d], f[1].u[d], f[2].u[d]);
    split_pack2(v[2], v[3], f[0].u[d + 1], f[1].u[d + 1], f[2].u[d + 1]);
}

// ReLU as ONE instruction: v_med3_f32(x, 0, 3e38) (with +inf the compiler folds it back to fmaxf, which adds a canonicalising
// v_max_f32 x, x in front under IEEE mode).  NOT inline asm: the hazard recogniser does not look into asm blocks, and a VALU read of an MFMA
// result needs software wait states on gfx950 — an asm v_max right behind the last MFMA read stale accumulators (r4: eval forward off by
// 1e-1 while the training forward, whose dropout hash happened to sit in between, was right).  A NaN activation comes out as 0 or 3e38
// here; NaN inputs are not a supported input.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f); }

// ---- weight image: one thread per (slice, out-feature row, lane group g) = 8 k-slots of the three planes
__global__ void __launch_bounds__(256) x6_prep_kernel(const float *__restrict__ P, int F, int NL, uint8_t *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n1 = x6_n1(F), ns = x6_nslices(F, NL);
    if (idx >= ns * kX6Rows * 4) return;
    const int g = idx & 3, row = (idx >> 2) % kX6Rows, sl = idx / (4 * kX6Rows);
    const int layer = sl < n1 ? 0 : 1 + (sl - n1) / 4, s = sl < n1 ? sl : (sl - n1) % 4;
    const int K = layer == 0 ? F : kH;
    const float *W = P + off_W(layer, F);
    Frag f[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = layer == 0 ? 32 * s + 8 * g + 4 * h + c : 32 * s + 16 * h + 4 * g + c;
            v[c] = (row < kH && k < K) ? W[(size_t)row * K + k] : 0.0f;
            if (layer > 0 && row < kH && k == kH) v[c] = P[off_b(layer, F) + row];      // the bias rides on the ones slot (feature 100) of B
        }
        split_pack4(v, f, 2 * h);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<u32x4 *>(img + (size_t)sl * kX6SliceBytes + (size_t)p * kX6PlaneBytes + (size_t)(row >> 4) * 1024 + g * 256 + (row & 15) * 16) = f[p].q;
}

using lds_u32x4 = __attribute__((address_space(3))) u32x4;
__device__ __forceinline__ uint32_t x6_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
// 64 lanes x 16 bytes, global (per-lane address) -> LDS (wave-uniform base + lane * 16); invisible to hipcc's waitcnt bookkeeping
__device__ __forceinline__ void x6_glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Raw buffer resource over [p, p + bytes): stores through it take a 32-bit byte offset and are DROPPED by the hardware when the offset
// lies outside — the activation stores of rows past R need no branch (a divergent `if` splits the MFMA regions into basic blocks)
using x6_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ x6_rsrc x6_srd(void *p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void x6_store16(x6_rsrc srd, uint32_t off, f32x4 v) {
#ifndef PTR_X6_STORE_AUX
#define PTR_X6_STORE_AUX 0
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srd, (int)off, 0, PTR_X6_STORE_AUX);
}
__device__ __forceinline__ f32x4 x6_load16(x6_rsrc srd, uint32_t voff, uint32_t soff) {      // out of range: zeros
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void x6_store4(x6_rsrc srd, uint32_t off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), srd, (int)off, 0, 0); }
constexpr uint32_t kX6Oob = 0xFFFFC000u;       // an offset no buffer of ours reaches — NOR the offset plus the largest immediate added to it
                                               // (6 x 1 KB feature tiles of a row tile: 0xFFFFF000 + 4096 wrapped to offset 0 and wrote row 0, r5)
constexpr uint64_t kX6BufLimit = 0xFFFFC000ull;

#ifndef PTR_X6_KTAIL
#define PTR_X6_KTAIL 1            /* r6: the last slice of a hidden layer as a 16-deep slice (0: the r4/r5 32-deep form, A/B measurements) */
#endif
#define X6_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).v, (B).v, (C), 0, 0, 0)

// =================================================================================================== forward
// LDS: ring kX6Ring x slice | biases NL x [112] | w_out [112] | b_out + pad
// STORE: write the post-dropout layer inputs `acts` (tile-major, ptr_mlp.h: [NL][ceil(R/16)][7][16][16]; column 100 of the first NL - 1 = 1.0)
//
// Schedule of one slice step of a wave (84 MFMAs: 7 out-feature tiles x 2 document tiles x 6 plane products), one region per tile:
//     read A(t1) | MFMA t0 + work 0 | read A(t2) | MFMA t1 + work 1 | read A(t3) | MFMA t2 + work 2 | SYNC | ... | read A(next slice, t0) | MFMA t6 + work 6
// A fragments are read one tile ahead into two register sets X / Y (an LDS read issued right in front of its MFMAs — what hipcc schedules
// on its own — exposes the LDS latency seven times per slice); seven tiles per slice swap the roles of the sets every slice (template
// parameter AP; layer 1 alternates, the four slices of a hidden layer restore it, an odd layer 1 pays one 12-register copy per tile).
// SYNC (s_waitcnt vmcnt + s_barrier in the MIDDLE of the step) publishes the NEXT slice, so the fragment stream never stops at a slice
// boundary, and hands the slot of the previous slice to the DMA.  "work" is the vector-ALU work that rides beside the MFMAs (probe,
// scratch/x6probe: up to two VALU instructions per v_mfma_f32_16x16x32_bf16 are free): in layer 1 the dropout + split of the NEXT slice's X
// fragment (and the loads of the slice behind it), in the last slice of a layer the epilogue (ReLU, dropout, store, split into the
// next layer's B fragments) of the tile that has just finished.  Registers: nothing may spill — a scratch reload waits with vmcnt for
// everything older, i.e. for the DMA pieces and X loads in flight (r4 trace: 9 K instead of 3.4 K cycles per step around the spills).
#define X6_SB() __builtin_amdgcn_sched_barrier(0)
#define X6_TSTRIDE 1024           /* bytes between the feature tiles of a row tile of the stored activations (tile-major, ptr_mlp.h) */
// interleave hint for the region in front of it: NM x (1 MFMA, then NV VALU)
#define X6_MIX(NM, NV)                                                                                     \
    do {                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) {                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);                                          \
        }                                                                                                  \
    } while (0)
template <int... Ks> struct X6K {};
template <int R, int K0, int... Ks> constexpr int x6_kget(X6K<K0, Ks...>) { if constexpr (R == 0) return K0; else return x6_kget<R - 1>(X6K<Ks...>{}); }

// DT: 16-document tiles per wave.  DT = 2: 8 waves per workgroup (two per SIMD, 256 registers each) — the form that is built.  DT = 4 (4 waves,
// one per SIMD, 512 registers: nothing spills, half the A-fragment LDS reads per document) was measured in r4 at 261 / 367 us against
// 240 / 352 us for DT = 2 (eval / training, 524288 x 136): no partner wave to hide a stall behind; its instantiation is not kept.
template <bool TRAIN, bool STORE, int DT>
__global__ void __launch_bounds__(16 / DT * 64, DT == 4 ? 1 : 2)
mlp_fwd_x6_kernel(const float *__restrict__ X, const float *__restrict__ P, const uint8_t *__restrict__ img, MlpArgs a,
                  float *__restrict__ preds, float *__restrict__ acts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_x6[];
    constexpr int NW = 16 / DT, RPT = 16 * DT, PW = (kX6Pieces + NW - 1) / NW;      // waves, rows per wave tile, DMA pieces per wave and slice
    const int F = a.F, NL = a.NL, R = a.R;
    float *Bs = reinterpret_cast<float *>(smem_x6 + (size_t)kX6Ring * kX6SliceBytes);
    float *Wo = Bs + (size_t)NL * kHP;
    const int tid = threadIdx.x;
    for (int i = tid; i < NL * kHP; i += NW * 64) { const int l = i / kHP, c = i - l * kHP; Bs[i] = c < kH ? P[off_b(l, F) + c] : 0.0f; }
    for (int i = tid; i < kHP + 16; i += NW * 64) Wo[i] = i < kH ? P[off_wout(NL, F) + i] : (i == kHP ? P[off_wout(NL, F) + kH] : 0.0f);
    __syncthreads();
    const float b_out = Wo[kHP];

    const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (R + RPT - 1) / RPT;
    const uint32_t thr = drop_thr(a.p_drop);
    const float scale = TRAIN ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int n1 = x6_n1(F), ns = n1 + 4 * (NL - 1);
    const uint32_t ring = x6_lds_addr(smem_x6);
    // this lane's 16 bytes of an A fragment (slice / plane / tile by offset): a tile is stored lane-linear — lane l = 16 g + j at byte 16 l —
    // which is what makes a ds_read_b128 conflict free (its 16-lane groups are {0-3, 12-15, 20-27}, ...: row-major [j][g] collides 2-way)
    const uint32_t lane_a = ring + (uint32_t)(lane * 16);

    // ---- the weight stream: slice n of the cyclic sequence 0 .. ns-1 lands in ring slot n mod kX6Ring.  Every wave issues three 1 KB
    // pieces per slice (21 pieces: the last waves repeat piece 20 — same bytes to the same place)
    int dma_slice = 0, dma_slot = 0;
    auto dma_issue = [&]() __attribute__((always_inline)) {
        const uint8_t *src = img + (size_t)dma_slice * kX6SliceBytes + lane * 16;
        const uint32_t dst = ring + (uint32_t)dma_slot * kX6SliceBytes;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int q = wave * PW + i;
            q = q < kX6Pieces ? q : kX6Pieces - 1;
            x6_glds16(src + q * 1024, __builtin_amdgcn_readfirstlane(dst + (uint32_t)q * 1024));
        }
        dma_slice = dma_slice + 1 == ns ? 0 : dma_slice + 1;
        dma_slot = dma_slot + 1 == kX6Ring ? 0 : dma_slot + 1;
    };
#pragma unroll 1
    for (int d = 0; d < kX6Ring - 1; ++d) dma_issue();
    static_assert(kX6Ring == 6 && (PW == 3 || PW == 6), "the vmcnt immediates below are PW * (kX6Ring - 2) and PW * (kX6Ring - 3) (+ the store margin)");
    if constexpr (STORE && DT == 2) {    // six more operations behind slices 1..4 (see slice_sync): slice 0 once more, into the free slot
#pragma unroll
        for (int i = 0; i < 6; ++i)
            x6_glds16(img + (size_t)i * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(ring + (uint32_t)(kX6Ring - 1) * kX6SliceBytes + (uint32_t)i * 1024));
    }
    // slice 0: this wave's pieces have landed when at most the pieces of slices 1..4 (+ the extra six) are outstanding (vmcnt counts in order)
    if constexpr (DT == 4) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    else if constexpr (STORE) asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    int cur_slot = 0;                // slot of the slice being multiplied
#ifdef PTR_X6_TRACE     // experiment builds: shader-clock stamps of workgroup 0 at every SYNC (arrival, release) behind the weight image
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(const_cast<uint8_t *>(img) + (size_t)ns * kX6SliceBytes) + wave * 256;
    int nstamp = 0;
#define X6_STAMP() do { if (blockIdx.x == 0 && lane == 0 && nstamp < 254) { trace[nstamp++] = clock64(); trace[nstamp & 1 ? 255 : 254] = wall_clock64(); if (nstamp == 1) trace[253] = wall_clock64(); } } while (0)   /* [253] / [255]: 100 MHz counter at the first / last stamp */
#else
#define X6_STAMP() do { } while (0)
#endif
    // SYNC in the middle of slice i: this wave's pieces of slice i + 1 have landed (behind them it has issued the 9 pieces of slices
    // i + 2 .. i + 4 at most); barrier: all of slice i + 1 is in LDS and every wave has finished slice i - 1, whose slot takes slice i + 5
    // Training: the activation stores count in vmcnt too (gfx9 has ONE in-order counter for loads and stores) and drain at the HBM write
    // rate — several steps per epilogue.  Behind the pieces of slice i + 1 a wave has issued, in ANY window of four steps, at least 14 more
    // vector-memory operations (the 14 stores of a layer's epilogue; in layer 1 the X loads: 4 per step), so 23 may stay outstanding
    // instead of 9 and the stores drain beside the MFMAs (the compiler's own scratch traffic would only add to the count).  The first
    // steps of a workgroup get the same margin from six extra DMA pieces issued behind the prologue's (into the slot slice 5 overwrites).
    auto slice_sync = [&]() __attribute__((always_inline)) -> uint32_t {
        X6_STAMP();
#ifdef PTR_X6_NOBAR          // ablation builds (wrong results, timing only)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        // DT = 4: 18 pieces behind slice i + 1, and in any four steps 28 stores (one epilogue) or 32 X loads; the prologue's 24 X loads
        // give the first steps the same margin
        if constexpr (DT == 4) { if constexpr (STORE) asm volatile("s_waitcnt vmcnt(46)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory"); }
        else if constexpr (STORE) asm volatile("s_waitcnt vmcnt(23)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
#endif
        X6_STAMP();
#ifndef PTR_X6_NODMA
        dma_issue();
#endif
        cur_slot = cur_slot + 1 == kX6Ring ? 0 : cur_slot + 1;
        return lane_a + (uint32_t)cur_slot * kX6SliceBytes;
    };

    // bias / w_out rows: ONE opaque base register each + immediate offsets (past the 64 KB `ds` offset range hipcc otherwise keeps a
    // loop-invariant address register per (tile) and spills them)
    using lds_f32x4 = __attribute__((address_space(3))) f32x4;
    uint32_t bs_base = x6_lds_addr(Bs) + (uint32_t)g * 16;
    asm volatile("" : "+v"(bs_base));
    uint32_t wo_delta = (uint32_t)NL * (kHP * 4);                // Wo = Bs + NL * 112 floats: a scalar added at the (seven) uses instead of a second register
    asm volatile("" : "+s"(wo_delta));
    auto lds4 = [](uint32_t base, int byte_off) __attribute__((always_inline)) { return *reinterpret_cast<lds_f32x4 *>((uintptr_t)(base + (uint32_t)byte_off)); };
    const uint32_t nrt = (uint32_t)act_row_tiles(R), layer_bytes = nrt * (uint32_t)(kActTile * 4);   // host: NL * layer_bytes < 2^32 - 4096
    const x6_rsrc xsrd = x6_srd(const_cast<float *>(X), (uint32_t)R * (uint32_t)(F * 4));     // host: R * F * 4 < 2^32 - 4096
    const x6_rsrc psrd = x6_srd(preds, (uint32_t)R * 4u);
    const x6_rsrc asrd = x6_srd(acts, STORE ? (uint32_t)NL * layer_bytes : 0u);
    f32x4 acc[kMT][DT];
    Frag bp[4][DT][3];               // B fragments of the current hidden layer: [slice][doc tile][plane]
    Frag bfx[2][DT][3];              // layer 1: B fragments of slice s in set s & 1
    f32x4 raw[2][DT][2];             // layer 1: X values of slice s in set s & 1, [doc tile][half] — loaded two steps before their fragments are built
    Frag af[2][3];                  // A fragments of one tile: [set][plane]
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    auto read_a = [&](Frag (&buf)[3], uint32_t abase, int mt) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) buf[p].q = *reinterpret_cast<lds_u32x4 *>((uintptr_t)(abase + (uint32_t)(p * kX6PlaneBytes + mt * 1024)));
    };
    // acc[mt][dt] += the six plane products of a slice, small terms first
    auto mma_tile = [&](const Frag (&buf)[3], auto mt_, const Frag (&bf)[DT][3]) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_)::value;
        constexpr int kA[6] = {0, 1, 2, 0, 1, 0}, kB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[mt][dt] = X6_MFMA(buf[kA[q]], bf[dt][kB[q]], acc[mt][dt]);
    };
    // r6: the LAST slice of a hidden layer holds only features 96..111 (100 real + the ones feature): its k slots (g, e >= 4) <-> features 112.. are zeros
    // in both operands, so the slice is 16 deep — one v_mfma_f32_16x16x16_bf16 on the first 8 bytes of each fragment (k slot (g, e < 4) <-> feature
    // 96 + 4 g + e) at HALF the matrix-pipe time of the 32-deep instruction that multiplied 16 zeros per lane.  Hidden layers: 3.5 instead of 4 slices.
    auto mma_tile_tail = [&](const Frag (&buf)[3], auto mt_, const Frag (&bf)[DT][3]) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_)::value;
        constexpr int kA[6] = {0, 1, 2, 0, 1, 0}, kB[6] = {2, 1, 0, 1, 0, 0};
        using i16x4 = __attribute__((ext_vector_type(4))) short;
        using u32x2_ = __attribute__((ext_vector_type(2))) uint32_t;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(i16x4, u32x2_{buf[kA[q]].u[0], buf[kA[q]].u[1]}),
                                                                         __builtin_bit_cast(i16x4, u32x2_{bf[dt][kB[q]].u[0], bf[dt][kB[q]].u[1]}), acc[mt][dt], 0, 0, 0);
    };
    // one slice step (see the schedule above).  In: af[AP] = tile 0 of this slice; out: af[AP ^ 1] = tile 0 of the next slice (its base
    // returned).  work(r): the VALU work beside the MFMAs of tile r; Ks: VALU instructions per MFMA of the interleave hint (0 = none)
    // LATE (the epilogue-carrying steps of the training kernel): the next tile's A fragments are read BEHIND the work of the current tile instead
    // of one tile ahead, so the read-ahead set is not live beside the epilogue's temporaries — the 12 registers that decide between spilling and
    // not spilling there (r5: a scratch reload waits with vmcnt for the activation stores in flight; the exposed LDS latency is hidden by the
    // partner wave, a drained store queue is not)
    auto slice_step = [&](auto ap_, uint32_t abase, const Frag (&bf)[DT][3], auto &&work, auto ks_, auto late_, auto tail_) __attribute__((always_inline)) -> uint32_t {
        constexpr int AP = decltype(ap_)::value;
        constexpr bool LATE = decltype(late_)::value;
        constexpr bool TAIL16 = decltype(tail_)::value;            // a 16-deep slice (the last one of a hidden layer)
        uint32_t nb = 0;
        static_for<kMT>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            constexpr int K = x6_kget<r>(decltype(ks_){});
            if constexpr (r == 3) nb = slice_sync();
            if constexpr (!LATE) {
                if constexpr (r < kMT - 1) read_a(af[AP ^ ((r + 1) & 1)], abase, r + 1);
                else read_a(af[AP ^ 1], nb, 0);
                X6_SB();
            }
            if constexpr (TAIL16) mma_tile_tail(af[AP ^ (r & 1)], r_, bf); else mma_tile(af[AP ^ (r & 1)], r_, bf);
            work(r_);
            if constexpr (K > 0) X6_MIX(6 * DT, K);
            X6_SB();
            if constexpr (LATE) {
                if constexpr (r < kMT - 1) read_a(af[AP ^ ((r + 1) & 1)], abase, r + 1);
                else read_a(af[AP ^ 1], nb, 0);
                X6_SB();
            }
        });
        return nb;
    };
#ifndef PTR_X6_LATE
#define PTR_X6_LATE 2
#endif
    using Early = std::false_type;
    using Full = std::false_type;
    using Tail16 = std::bool_constant<PTR_X6_KTAIL != 0>;
    using LateE = std::bool_constant<TRAIN && PTR_X6_LATE != 0>;      // steps whose work is an epilogue
    using KNone = X6K<0, 0, 0, 0, 0, 0, 0>;
    auto nowork = [](auto) __attribute__((always_inline)) {};

    // tiles of 32 documents: workgroup b, wave w, pass it -> tile (it * gridDim.x + b) * 8 + w; every wave of a workgroup runs the same
    // number of passes (lockstep), a wave whose tile lies past the end runs masked (clamped loads, no stores)
    const int npass = (ntiles + NW * (int)gridDim.x - 1) / (NW * (int)gridDim.x);
    auto tile_of = [&](int it) __attribute__((always_inline)) { return (it * (int)gridDim.x + (int)blockIdx.x) * NW + wave; };
    // Per-lane constants, one register each; everything that depends on the tile is a SCALAR added per use (s_mul / s_add on the scalar
    // unit, one v_add or an soffset operand on the vector side) — per-tile VGPR copies of rows, offsets and keys cost a dozen registers
    // this kernel does not have:
    //   jX / l16  byte offset of the lane's 32 bytes inside row j of X / of its 16 bytes inside a (16 rows x 16 features) block of `acts`
    //             (row j at 64 j, its features 4 g .. + 3 at + 16 g: the block is one contiguous KB per store instruction either way)
    //   jH / jK   dropout key parts  j * kDropRowMul + g * kDropFgMul + seed_lo (hidden sites, fg = 4 mt + g) / with 2 g (X site, fg = 8 s + 2 g + h)
#ifdef PTR_X6_LANE_LINEAR   // timing experiment (wrong block layout): lane l at byte 16 l of its (16 rows x 16 features) block
    uint32_t jX = (uint32_t)j * (uint32_t)(F * 4) + (uint32_t)g * 32, l16 = (uint32_t)lane * 16;
#else
    uint32_t jX = (uint32_t)j * (uint32_t)(F * 4) + (uint32_t)g * 32, l16 = (uint32_t)(j * 64 + g * 16);   // row j, features 4 g .. + 3 of the block
#endif
    uint32_t jH = (uint32_t)j * kDropRowMul + (uint32_t)g * kDropFgMul + a.seed_lo, jK = jH + (uint32_t)g * kDropFgMul;
    asm volatile("" : "+v"(jX), "+v"(l16), "+v"(jH), "+v"(jK));
    // quarter q = (dt, h) of slice s of the 32 documents from row t32 on: the float4 a lane contributes to the B fragment.  X is read through
    // a buffer resource: rows past R and columns past F get an out-of-range offset (zeros)
    auto load_xq = [&](f32x4 (&rw)[DT][2], int t32, int s, int q) __attribute__((always_inline)) {
        const int dt = q >> 1, h = q & 1;
#ifdef PTR_X6_NOX
        const uint32_t vo = jX, so = (uint32_t)((t32 + 16 * dt) & 1023) * (uint32_t)(F * 4);
#else
        // columns past F (the last slice when F % 32 != 0) are out of range like the rows past R: they read zeros, NOT the next row's first
        // features — a NaN / Inf there would reach this row's products as 0 * Inf
        // (the column test on the SCALAR unit: lane groups g < ng hold columns below F — a 64-bit lane mask ANDed with the row test's,
        // one v_cndmask; a per-lane compare costs a register the training kernel does not have)
        const int left = F * 4 - 128 * s - 16 * h;                                     // bytes of the row from this quarter's column of group 0 on
        const int ng = left <= 0 ? 0 : (left + 31) >> 5;                               // lane groups with a column below F
        const uint64_t cmask = ng >= 4 ? ~0ull : ((1ull << (16 * ng)) - 1ull);
        const uint64_t m = __builtin_amdgcn_ballot_w64(j < R - t32 - 16 * dt) & cmask;
        const uint32_t vo = __builtin_amdgcn_inverse_ballot_w64(m) ? jX : kX6Oob;
        const uint32_t so = (uint32_t)(t32 + 16 * dt) * (uint32_t)(F * 4);
#endif
        rw[dt][h] = x6_load16(xsrd, vo, so + (uint32_t)(128 * s + 16 * h));
    };
    // quarter q of the X fragment of slice s: input dropout (site 0, fg = 8 s + 2 g + h) + split
    auto make_bq = [&](const f32x4 (&rw)[DT][2], int t32, int s, Frag (&bf)[DT][3], int q) __attribute__((always_inline)) {
        const int dt = q >> 1, h = q & 1;
        f32x4 v = rw[dt][h];
        if constexpr (TRAIN) {
            uint32_t w0, w1;
            drop_bits_key(jK + ((uint32_t)(t32 + 16 * dt) * kDropRowMul + (uint32_t)(8 * s + h) * kDropFgMul), a.seed_hi, w0, w1);
            v = drop4(v, w0, w1, thr, scale);
        }
        split_pack4(v, bf[dt], 2 * h);
    };
    auto bias_init = [&]() __attribute__((always_inline)) {            // layer 1's bias
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            const f32x4 b4 = lds4(bs_base, 64 * mt);
            acc[mt][0] = b4; acc[mt][1] = b4;
        }
    };

    // kernel prologue: the fragments of slice 0 of the first tile, the X values of slice 1, the first A fragment
    {
        const int t0 = tile_of(0) * RPT;
#pragma unroll
        for (int q = 0; q < 2 * DT; ++q) { load_xq(raw[0], t0, 0, q); load_xq(raw[1], t0, n1 > 1 ? 1 : 0, q); }
#pragma unroll
        for (int q = 0; q < 2 * DT; ++q) { make_bq(raw[0], t0, 0, bfx[0], q); load_xq(raw[0], t0, n1 > 2 ? 2 : 0, q); }
    }
    bias_init();
    uint32_t abase = lane_a;
    read_a(af[0], abase, 0);
    // VALU per MFMA of the hints: an X-fragment quarter is ~22 (eval) / ~50 (training) instructions, a tile's epilogue ~52 / ~110
#ifndef PTR_X6_KE_TRAIN
#define PTR_X6_KE_TRAIN 9
#endif
#ifndef PTR_X6_KX_TRAIN
#define PTR_X6_KX_TRAIN 4
#endif
    constexpr int KX = TRAIN ? PTR_X6_KX_TRAIN : 2, KE = TRAIN ? PTR_X6_KE_TRAIN : 5;

#pragma unroll 1
    for (int it = 0; it < npass; ++it) {
        const int tile = tile_of(it), tile_next = tile_of(it + 1);
        const int t32 = tile * RPT, t32n = tile_next * RPT;        // first row of this / the next tile
        auto store_off = [&](int layer, int dt) __attribute__((always_inline)) {       // opaque: `+ 64 mt` stays an immediate instead of seven hoisted select operands
            // tile-major: this wave's 16-row tile is 7 contiguous KB, a store instruction = lane * 16 + 1024 * feature tile.  Whole tiles are
            // written (rows past R of the last tile hold finite values nobody uses); tiles past the last one get an out-of-range offset
            const uint32_t rt = (uint32_t)(t32 + 16 * dt) >> 4;
#ifdef PTR_X6_L2STORE      // ablation: the same store instructions, aimed at an L2-resident part of the buffer
            uint32_t o = rt < nrt ? l16 + ((rt & 127u) * (uint32_t)(kActTile * 4)) : kX6Oob;
#else
            uint32_t o = rt < nrt ? l16 + (rt * (uint32_t)(kActTile * 4) + (uint32_t)layer * layer_bytes) : kX6Oob;
#endif
            asm volatile("" : "+v"(o));
            return o;
        };
        auto epilogue = [&](auto mt_, int site) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_)::value;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 h = acc[mt][dt];
                acc[mt][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};        // the next layer's first MFMA takes the inline constant: no live registers
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = relu1(h[c]);
                if constexpr (TRAIN) {
                    uint32_t w0, w1;
                    drop_bits_key(jH + ((uint32_t)(t32 + 16 * dt) * kDropRowMul + (uint32_t)(4 * mt) * kDropFgMul + (uint32_t)site * kDropSiteMul), a.seed_hi, w0, w1);
                    h = drop4(h, w0, w1, thr, scale);
                }
                if constexpr (STORE) {
                    f32x4 o = h;
                    if (mt == kMT - 1 && g == 1) o[0] = 1.0f;              // the ones column (100) of the fused backward
#ifndef PTR_X6_NOSTORE
                    x6_store16(asrd, store_off(site - 1, dt) + X6_TSTRIDE * mt, o);
#endif
                }
                split_pack4(h, bp[mt >> 1][dt], 2 * (mt & 1));
#ifdef PTR_X6_EPI_SB
                if constexpr (TRAIN) X6_SB();
#endif
                if (mt == kMT - 1) {
#if !PTR_X6_KTAIL       /* (the 16-deep tail slice never reads the second half of these fragments) */
#pragma unroll
                    for (int p = 0; p < 3; ++p) { bp[3][dt][p].u[2] = 0u; bp[3][dt][p].u[3] = 0u; }
#endif
                    // feature 100 (lane group 1, element 0; its activation is exactly 0) = 1.0: the slot the weight image keeps the bias in
                    bp[3][dt][0].u[0] |= g == 1 ? 0x3F80u : 0u;
                }
            }
        };
        // last hidden activation + output layer (100 -> 1): VALU dot product per lane, reduced over the 4 lane groups at the end;
        // the accumulators restart from the bias of layer 1
        float sc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) sc[dt] = 0.0f;
        auto epilogue_out = [&](auto mt_) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_)::value;
            const f32x4 w4 = lds4(bs_base + wo_delta, 64 * mt);
            const f32x4 bn = lds4(bs_base, 64 * mt);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4 h = acc[mt][dt];
                acc[mt][dt] = bn;
#pragma unroll
                for (int c = 0; c < 4; ++c) { h[c] = relu1(h[c]); sc[dt] = fmaf(h[c], w4[c], sc[dt]); }
                if constexpr (STORE) {
#ifndef PTR_X6_NOSTORE
                    x6_store16(asrd, store_off(NL - 1, dt) + X6_TSTRIDE * mt, h);
#endif
                }
            }
        };

        // ---- layer 1.  Step s multiplies bfx[s & 1] and, beside the MFMAs of tiles 0 / 1 / 3 / 4, builds quarter q of the fragments of
        // slice s + 1 from raw[(s + 1) & 1] into bfx[(s + 1) & 1], reloading the quarter with slice s + 3 (past the end: a valid address,
        // never used).  Two steps of distance: in training the loads queue behind the activation stores of the last epilogue (vmcnt is
        // one in-order counter), and a load consumed one step after a 14-store burst stalled the whole lockstep workgroup (r4 trace)
        auto l1_step = [&](auto par_, int s) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_)::value;
            abase = slice_step(par_, abase, bfx[PAR], [&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                auto quarter = [&](int q) __attribute__((always_inline)) { make_bq(raw[PAR ^ 1], t32, s + 1, bfx[PAR ^ 1], q); load_xq(raw[PAR ^ 1], t32, s + 3 < n1 ? s + 3 : 0, q); };
                if constexpr (DT == 2) {            // quarters 0..3 beside tiles 0, 1, 3, 4
                    constexpr int q = r == 0 ? 0 : r == 1 ? 1 : r == 3 ? 2 : r == 4 ? 3 : -1;
                    if constexpr (q >= 0) quarter(q);
                } else {                            // eight quarters: two beside tile 0, one beside each other tile
                    if constexpr (r == 0) { quarter(0); quarter(1); } else quarter(r + 1);
                }
            }, std::conditional_t<DT == 2, X6K<KX, KX, 0, KX, KX, 0, 0>, X6K<KX, KX / 2, KX / 2, KX / 2, KX / 2, KX / 2, KX / 2>>{}, Early{}, Full{});
        };
        // the last slice of layer 1: the epilogue of a tile rides beside the MFMAs of the next one
        auto l1_last = [&](auto par_) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_)::value;
            abase = slice_step(par_, abase, bfx[PAR], [&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                if constexpr (r > 0) epilogue(std::integral_constant<int, r - 1>{}, 1);
            }, X6K<0, KE, KE, KE, KE, KE, KE>{}, LateE{}, Full{});
            epilogue(std::integral_constant<int, kMT - 1>{}, 1);
            if constexpr (PAR == 0) {               // an odd number of layer-1 slices: the hidden layers expect the next A fragment in set 0
#pragma unroll
                for (int p = 0; p < 3; ++p) af[0][p].q = af[1][p].q;
            }
        };
        // (the odd set is only ever read right behind the even step that fills it — spelled out in the control flow, or bfx[1] stays live
        // through the hidden layers for the register allocator)
        int s = 0;
        for (; s + 2 <= n1 - 1; s += 2) { l1_step(I0{}, s); l1_step(I1{}, s + 1); }
        if (s < n1 - 1) { l1_step(I0{}, s); l1_last(I1{}); } else { l1_last(I0{}); }

        // ---- hidden layers 2 .. NL: B fragments = the registers the previous epilogue left
        auto hidden = [&](auto lastlayer_, int l) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(lastlayer_)::value;
            abase = slice_step(I0{}, abase, bp[0], nowork, KNone{}, Early{}, Full{});
            abase = slice_step(I1{}, abase, bp[1], nowork, KNone{}, Early{}, Full{});
#ifndef PTR_X6_XLOAD_POS
#define PTR_X6_XLOAD_POS 0          /* where the next tile's first X loads are issued: 0 behind the layer's second slice step (r4/r5), 1 behind its third, 2 inside its third (behind the SYNC) */
#endif
#ifndef PTR_X6_XQ_R0
#define PTR_X6_XQ_R0 0              /* first tile region of the last step that converts a quarter of the next tile's slice 0 */
#endif
            auto xloads = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 2 * DT; ++q) { load_xq(raw[0], t32n, 0, q); load_xq(raw[1], t32n, n1 > 1 ? 1 : 0, q); }      // (past the end: zeros, never used)
            };
            if constexpr (LAST && PTR_X6_XLOAD_POS == 0) {       // the next tile's first two X slices, in front of this layer's store burst
                xloads();
                X6_SB();
            }
            if constexpr (LAST && PTR_X6_XLOAD_POS == 2) {
                abase = slice_step(I0{}, abase, bp[2], [&](auto r_) __attribute__((always_inline)) { if constexpr (decltype(r_)::value == 4) xloads(); }, KNone{}, Early{}, Full{});
            } else {
                abase = slice_step(I0{}, abase, bp[2], nowork, KNone{}, Early{}, Full{});
            }
            if constexpr (LAST && PTR_X6_XLOAD_POS == 1) {
                xloads();
                X6_SB();
            }
            if constexpr (LAST) {
                abase = slice_step(I1{}, abase, bp[3], [&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    // the third slice right behind the quarter it replaces: in front of all but the first few stores of the burst
                    auto quarter = [&](int q) __attribute__((always_inline)) { make_bq(raw[0], t32n, 0, bfx[0], q); load_xq(raw[0], t32n, n1 > 2 ? 2 : 0, q); };
                    if constexpr (DT == 2) { if constexpr (r >= PTR_X6_XQ_R0 && r < PTR_X6_XQ_R0 + 4) quarter(r - PTR_X6_XQ_R0); }
                    else if constexpr (r < 4) { quarter(2 * r); quarter(2 * r + 1); }
                    if constexpr (r > 0) epilogue_out(std::integral_constant<int, r - 1>{});
                }, X6K<KX, KX + 1, KX + 1, KX + 1, 1, 1, 1>{}, std::bool_constant<TRAIN && PTR_X6_LATE == 2>{}, Tail16{});
                epilogue_out(std::integral_constant<int, kMT - 1>{});
            } else {
                abase = slice_step(I1{}, abase, bp[3], [&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    if constexpr (r > 0) epilogue(std::integral_constant<int, r - 1>{}, l + 1);
                }, X6K<0, KE, KE, KE, KE, KE, KE>{}, LateE{}, Tail16{});
                epilogue(std::integral_constant<int, kMT - 1>{}, l + 1);
            }
        };
#pragma unroll 1
        for (int l = 1; l < NL - 1; ++l) hidden(std::false_type{}, l);
        hidden(std::true_type{}, NL - 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float t = sc[dt];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            x6_store4(psrd, (g == 0 && j < R - t32 - 16 * dt) ? (uint32_t)(t32 + 16 * dt + j) * 4u : kX6Oob, t + b_out);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead DMA must not outlive the workgroup's LDS allocation
}

bool x6_supported(int F, int NL) { return F % 4 == 0 && F >= 4 && NL >= 2 && NL <= kMaxLayers && x6_lds_bytes(NL) <= 160 * 1024; }

}  // namespace ptr

extern "C" size_t ptr_mlp_x6_ws_bytes(int F, int NL) {
    return ptr::x6_supported(F, NL) ? (size_t)ptr::x6_nslices(F, NL) * ptr::kX6SliceBytes + 16384 : 0;     // + 16 KB: stamps of trace builds
}

extern "C" int ptr_mlp_forward_x6(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed,
                                  float *preds, float *acts, void *wimg, void *stream) {
    return ptr::mlp_forward_x6_impl(X, params, R, F, NL, train, p_drop, seed, preds, acts, wimg, true, stream);
}

int ptr::mlp_forward_x6_impl(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed, float *preds, float *acts,
                             void *wimg, bool prep, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mlp_forward_x6";
    if (R < 0 || F <= 0 || NL < 1 || NL > kMaxLayers) { set_error("%s: bad shape R=%d F=%d NL=%d", who, R, F, NL); return PTR_ERR_INVALID_ARG; }
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) { set_error("%s: dropout p=%g out of [0,1)", who, (double)p_drop); return PTR_ERR_INVALID_ARG; }
    if (!x6_supported(F, NL)) { set_error("%s: F=%d NL=%d is outside the bf16x6 scorer's range (F %% 4 == 0, 2 <= NL <= 8)", who, F, NL); return PTR_ERR_UNSUPPORTED; }
    if (R > 0 && (!X || !params || !preds || !wimg || (train && !acts))) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(wimg) & 15) || (acts && (reinterpret_cast<uintptr_t>(acts) & 15))) {
        set_error("%s: X, acts and wimg must be 16-byte aligned", who); return PTR_ERR_INVALID_ARG;
    }
    if (R == 0) return 0;
    if ((uint64_t)R * (uint64_t)F * 4 >= kX6BufLimit) {
        set_error("%s: R * F * 4 bytes of X exceed the 4 GB a buffer resource addresses (R=%d, F=%d): split the batch", who, R, F);
        return PTR_ERR_UNSUPPORTED;
    }
    if (train && (uint64_t)NL * (uint64_t)act_layer_floats(R) * 4 >= kX6BufLimit) {
        set_error("%s: NL * ceil16(R) * 448 bytes of activations exceed the 4 GB a buffer resource addresses (R=%d): split the batch", who, R);
        return PTR_ERR_UNSUPPORTED;
    }
    MlpArgs a{R, F, NL, train ? p_drop : 0.0f, (uint32_t)seed, (uint32_t)(seed >> 32)};
    hipStream_t st = as_stream(stream);
    if (prep) {                       // (prep = false: the previous train step's optimiser launch left the image current, ptr_train_step)
        const int nthreads = x6_nslices(F, NL) * kX6Rows * 4;
        hipLaunchKernelGGL(x6_prep_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, st, params, F, NL, reinterpret_cast<uint8_t *>(wimg));
        if (int e = check_hip(hipGetLastError(), who)) return e;
    }
    const size_t lds = x6_lds_bytes(NL);
    constexpr int dt = 2, nw = 16 / dt, rpt = 16 * dt;
    const int ntiles = (R + rpt - 1) / rpt, nblk = (ntiles + nw - 1) / nw;
    const int grid = nblk < mlp_num_cus() ? nblk : mlp_num_cus();
    auto launch = [&](auto kern) -> int {
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, st, X, params, reinterpret_cast<const uint8_t *>(wimg), a, preds, acts);
        return check_hip(hipGetLastError(), who);
    };
    return train ? launch(mlp_fwd_x6_kernel<true, true, 2>) : launch(mlp_fwd_x6_kernel<false, false, 2>);
}
