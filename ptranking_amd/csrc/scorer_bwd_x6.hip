// Single-pass fused backward of the pointsf scorer on bf16 matrix instructions with fp32 results ("bf16 x 6", see scorer_x6.hip for the
// arithmetic): dZ chain + every weight gradient in ONE kernel that reads X and the stored activations once and never materialises dZ in HBM.
//
// Reference: the autograd backward of ptranking/base/point_ranker.py:45-55 + ptranking/base/utils.py:288-356
// ((Dropout -> Linear -> ReLU) x 3 -> Linear).  Served: three hidden layers, 129 <= F <= 143, F % 4 == 0 (the shapes of the fp32-MFMA fused
// backward, scorer_bwd.hip, whose partial-gradient layout and reduction it shares); everything else takes the fp32 kernels.
//
// Structure.  A persistent 8-wave workgroup per CU walks slabs of 32 documents (two 16-document tiles = one 32-deep contraction slice of
// the weight gradients).  Per slab the operands live in LDS as bf16 PLANE IMAGES [3 planes][32 documents][features] (row stride 224 B for
// 112 features, 288 B for the 144 of X — strides ds_read_b64_tr_b16 reads without bank conflicts, scratch/x6probe):
//     ZA, ZB   dZ of the hidden layers (dZ3 -> ZA, dZ2 -> ZB, dZ1 -> ZA again)
//     A1, A2   the stored activations of layers 1 / 2 (column 100 = the ones column the forward stores: column 100 of a dW tile row is db)
//     XI       the input features with the input dropout recomputed (column F = 1: db of layer 1)
// and are consumed by v_mfma_f32_16x16x32_bf16 in two roles:
//   * dZ chain ("transposed world"): dA^T[in][doc] = W^T[in][out] * dZ^T[out][doc].  Wave w (0..6) owns in-feature tile w of both chain
//     layers and keeps its W^T fragments — split into planes once, in the prologue — in REGISTERS for the whole kernel (2 layers x 4 slices
//     x 3 planes); the B operand is a ds_read_b128 of a dZ image (8 consecutive out-features of one document), the result is gated by the
//     stored activation (a 4-bit mask kept from the staging pass), split and written back as the next dZ image.
//   * weight gradients ("document-contraction world"): dW_l[out][in] = sum_docs dZ_l[doc][out] * A_{l-1}[doc][in] with the slab's 32 documents
//     as the k index: both operands come out of the [doc][feature] images through ds_read_b64_tr_b16, the transpose-read that hands lane
//     (feature, group G) the documents 4G..4G+3 (+16) of its feature.  Wave w owns out-feature row w of dW_3, dW_2 and in-tiles 0..3 of
//     dW_1; wave 7 (no chain tile) owns in-tiles 4..8 of dW_1 for all seven rows: 18 / 35 accumulator tiles per wave, all in registers.
// The stored activations of the NEXT slab land in an fp32 staging area by LDS-DMA while the current slab is multiplied; X and dLoss/dscore
// are prefetched into registers.  Four barriers per slab: staging pass | chain 3 + dW_3 | chain 2 + dW_2 | dW_1.
// Results are bit-stable: fixed tile ownership, fixed document order, per-workgroup partials reduced by reduce_partials_kernel.
#include "ptr_mlp.h"

namespace ptr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
union BFrag { bf16x8 v; u32x4 q; uint32_t u[4]; };

constexpr int kB6S = 32;                        // documents per slab
constexpr int kB6ZRS = 224, kB6ZPL = kB6S * kB6ZRS, kB6ZIMG = 3 * kB6ZPL;        // 112-column images: row stride, plane, image (21504 B)
constexpr int kB6XRS = 288, kB6XPL = kB6S * kB6XRS, kB6XIMG = 3 * kB6XPL;        // the X image: 144 columns (27648 B)
constexpr int kB6STG = kB6S * kAL * 4;                                           // fp32 staging of one layer's slab (14336 B)
constexpr int kB6_ZA = 0, kB6_ZB = kB6ZIMG, kB6_A1 = 2 * kB6ZIMG, kB6_A2 = 3 * kB6ZIMG, kB6_XI = 4 * kB6ZIMG, kB6_ST = kB6_XI + kB6XIMG;
constexpr int kB6_WO = kB6_ST + 3 * kB6STG;                                      // w_out [112] (re-read per slab: four registers less)
constexpr int kB6Lds = kB6_WO + 512;
static_assert(kB6Lds <= 160 * 1024, "LDS budget");

__device__ __forceinline__ uint32_t b6_cvt_pk(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2)); }
// two fp32 values -> one dword of each plane (round-to-nearest split, see scorer_x6.hip split_pack2)
__device__ __forceinline__ void b6_split2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = b6_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = b6_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = b6_cvt_pk(s0, s1);
}
using lds_u32x4_b = __attribute__((address_space(3))) u32x4;
using lds_u32x2_b = __attribute__((address_space(3))) u32x2;
using lds_f32x4_b = __attribute__((address_space(3))) f32x4;
using lds_i16x4_b = __attribute__((address_space(3))) i16x4;
__device__ __forceinline__ uint32_t b6_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
// Global memory is read through buffer resources (32-bit lane offset + scalar offset; out-of-range lanes read zeros): 64-bit per-lane
// pointers cost two registers per (piece, tile) once hipcc hoists their loop-invariant parts.
using b6_srd = __attribute__((ext_vector_type(4))) int;
using b6_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ b6_srd b6_make_srd(const void *p, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    return b6_srd{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 bytes, buffer (lane offset + scalar offset) -> LDS (wave-uniform base + lane * 16); invisible to hipcc's waitcnt bookkeeping
__device__ __forceinline__ void b6_bdma16(b6_srd srd, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst) : "memory");
}
// four consecutive features of one document -> the three planes of an image (8 bytes each)
__device__ __forceinline__ void b6_write4(uint32_t addr, int plane_bytes, const f32x4 v) {
    uint32_t a[3], b[3];
#ifdef PTR_B6_ABL_NOVALU     // timing-only ablation (wrong results): no splitting
    a[0] = a[1] = a[2] = __float_as_uint(v[0]); b[0] = b[1] = b[2] = __float_as_uint(v[2]);
#else
    b6_split2(v[0], v[1], a[0], a[1], a[2]);
    b6_split2(v[2], v[3], b[0], b[1], b[2]);
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<lds_u32x2_b *>((uintptr_t)(addr + (uint32_t)(p * plane_bytes))) = u32x2{a[p], b[p]};
}
// LDS writes of this wave complete (lgkmcnt) -> workgroup barrier; NOT __syncthreads(): its fence would also drain the DMA / prefetch loads
__device__ __forceinline__ void b6_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// A scalar the compiler cannot fold: image bases go through this INSIDE the slab loop, or hipcc precomputes one address register per (image,
// plane, tile) combination — the images lie past the 64 KB ds offset range — hoists the dozens of them out of the loop and spills them
__device__ __forceinline__ uint32_t b6_opaque(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
#ifndef PTR_B6_DMA_CHAIN_ONLY
#define PTR_B6_DMA_CHAIN_ONLY 1
#endif
#ifndef PTR_B6_PREFETCH_LATE
#define PTR_B6_PREFETCH_LATE 1        /* 1: the next-next slab's DMA is issued at the end of the chain-3 phase instead of right behind B4 */
#endif
#ifndef PTR_B6_PIPE_W7
#define PTR_B6_PIPE_W7 1
#endif
#ifndef PTR_B6_PIPE_C2
#define PTR_B6_PIPE_C2 0
#endif
#ifndef PTR_B6_PRIO
#define PTR_B6_PRIO 0
#endif
#ifndef PTR_B6_X_PHASE
#define PTR_B6_X_PHASE 1              /* chain phase (0 / 1) in which X is loaded and turned into its plane image */
#endif
#ifndef PTR_B6_SWZ
#define PTR_B6_SWZ 1
#endif
#ifndef PTR_B6_STAGE_ORDER
#define PTR_B6_STAGE_ORDER 0
#endif
#ifndef PTR_B6_TAILPRE
#define PTR_B6_TAILPRE 1              /* r6: the chain's 16-deep tail operands are read beside the last full slice's MFMAs (in the registers the B-fragment prefetch no longer needs) */
#endif
#ifndef PTR_B6_CHAIN_ORDER
#define PTR_B6_CHAIN_ORDER 0
#endif
#ifndef PTR_B6_EPI_MIX
#define PTR_B6_EPI_MIX 0              /* r6 experiment: the chain's epilogue (gating, split, image stores) issued BETWEEN the dW row's MFMAs (2-3 VALU per MFMA) instead of in front of them */
#endif
#ifndef PTR_B6_EPI_K
#define PTR_B6_EPI_K 3
#endif
#ifndef PTR_B6_W7_ILV
#define PTR_B6_W7_ILV 0               /* experiment: wave 7's two tiles of a row as one interleaved stream (no back-to-back MFMAs on one accumulator) */
#endif
#ifndef PTR_B6_W7_DEPTH
#define PTR_B6_W7_DEPTH 2             /* experiment: dZ tiles in flight for wave 7 (2 = one row ahead, 3 = two rows ahead) */
#endif
#ifndef PTR_B6_W7HOLD
#define PTR_B6_W7HOLD 1               /* r6: wave 7 keeps the in-tile fragments its rows share in registers for the phase and streams only the dZ tiles */
#endif
#define B6_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).v, (B).v, (C), 0, 0, 0)

template <int NT1>
__global__ void __launch_bounds__(512, 2)
mlp_bwd_x6_kernel(const float *__restrict__ X, const float *__restrict__ P, const float *__restrict__ acts, const float *__restrict__ dpreds,
                  MlpArgs a, float *__restrict__ ws, size_t np_stride, float *__restrict__ dz0) {
    static_assert(NT1 == 9 || NT1 == 0, "the X image is laid out for nine in-feature tiles; NT1 = 0 is the TAIL form");
    // NT1 == 0: the TAIL form for every other input width (r5; the fp32-MFMA kernel's TAIL form, scorer_bwd.hip, on the bf16 instructions): no X
    // image, no dW_1 tiles — the chain's last result dZ_1 goes to HBM as fp32 (dz0 [R][112], row-major) for the first-layer dW kernel, which also takes
    // db_1 off its column sums; what is left of the third phase is the next slab's staging pass.
    constexpr bool TAIL = NT1 == 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_b6[];
    constexpr int NL = 3;
    const int F = a.F, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, W = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = b6_lds_addr(smem_b6);
#ifdef PTR_B6_TRACE_EDGE      // experiment builds: shader-clock stamps of the kernel's prologue / epilogue (workgroup 0, every wave), behind the phase stamps' area
    unsigned long long *etrace = reinterpret_cast<unsigned long long *>(ws + (size_t)gridDim.x * np_stride) + 8 * 256 + W * 16;
    int nedge = 0;
#define B6_EDGE() do { if (blockIdx.x == 0 && lane == 0 && nedge < 16) etrace[nedge++] = clock64(); } while (0)
#else
#define B6_EDGE() do { } while (0)
#endif
    B6_EDGE();                                                       // E0 entry
    const int nslabs = (R + kB6S - 1) / kB6S;
    const uint32_t thr = a.p_drop > 0.0f ? drop_thr(a.p_drop) : 0u;
    const float scale = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const bool chain = W < 7;
    // every image starts as zeros (rows of documents past R in the last slab are read before anything was written there: 0 x NaN = NaN)
    for (int i = tid; i < kB6Lds / 16; i += 512) reinterpret_cast<u32x4 *>(smem_b6)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();                       // before the first DMA lands in the staging area
    B6_EDGE();                                                       // E1 LDS zeroed

    // ---- persistent registers: ONE array of 42 x 4, used by role (a single instruction stream allocates the maximum over the waves anyway —
    // separate arrays for the chain fragments and the accumulators would add up: 296 spills in the first build):
    //   chain wave w: st[0..4] dW_3 row w in-tiles 0..4, st[5..9] dW_2 row w in-tiles 0..4, st[10..17] dW_1 row w in-tiles 0..7 (18 accumulator
    //                 tiles); its W^T fragments as bits: st[18 + 3 (3 c + s) + p] = plane p of the full slice s < 3 of chain layer c, and the
    //                 contraction TAIL (out-features 96..99 + the zero padding to 111) as 16-deep fragments, two dwords per plane: dword
    //                 2 (3 c + p) + {0, 1} of st[36..38] — 21 fragment slots instead of 24, and no B fragment reads past its image row
    //   wave 7:       st[2 m + q] dW_3 row m in-tile 5 + q, st[14 + 2 m + q] the same of dW_2, st[28 + m] dW_1 row m in-tile 8 (35 tiles)
    // so that every phase is balanced: 48 chain + 30 dW MFMAs per chain wave against 84 dW MFMAs of wave 7 in the two chain phases, 48 / 42 in
    // the last one.
    f32x4 st[39];
#pragma unroll
    for (int i = 0; i < 39; ++i) st[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (chain) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {              // chain c = 0: layer 2 (dZ3 -> dA2), c = 1: layer 1 (dZ2 -> dA1)
            const float *Wl = P + off_W(2 - c, F);
            const int ri = 16 * W + j;                              // in-feature = row of W^T
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 32 * s + 8 * g + e;               // out-feature = contraction index (< 96)
                    v[e] = ri < kH ? Wl[(size_t)k * kH + ri] : 0.0f;
                }
                BFrag pl[3];
#pragma unroll
                for (int d = 0; d < 4; ++d) b6_split2(v[2 * d], v[2 * d + 1], pl[0].u[d], pl[1].u[d], pl[2].u[d]);
#pragma unroll
                for (int p = 0; p < 3; ++p) st[18 + 3 * (3 * c + s) + p] = __builtin_bit_cast(f32x4, pl[p].q);
            }
            {   // tail: k = 96 + 4 g + e, e < 4
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int k = 96 + 4 * g + e; v[e] = (ri < kH && k < kH) ? Wl[(size_t)k * kH + ri] : 0.0f; }
                uint32_t t0[3], t1[3];
                b6_split2(v[0], v[1], t0[0], t0[1], t0[2]);
                b6_split2(v[2], v[3], t1[0], t1[1], t1[2]);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int d = 2 * (3 * c + p);                  // dword index among the 12 tail dwords (st[36..38])
                    st[36 + d / 4][d % 4] = __uint_as_float(t0[p]);
                    st[36 + d / 4][d % 4 + 1] = __uint_as_float(t1[p]);
                }
            }
        }
    }
    B6_EDGE();                                                       // E2 W^T fragments loaded and split
    if (tid < kHP) reinterpret_cast<float *>(smem_b6 + kB6_WO)[tid] = tid < kH ? P[off_wout(NL, F) + tid] : 0.0f;
    float awo[4] = {0.0f, 0.0f, 0.0f, 0.0f}, abo = 0.0f;

    // per-lane addresses
    // r5: the 8-byte chunks of a row's 32-byte feature-tile groups are XOR-swizzled by a row bit (chunk g of row r sits at g ^ 2 ((r >> 2) & 1)): a plane-image
    // store (ds_write_b64: 16 consecutive lanes = the 16 rows of one chunk column, 224-byte stride = 4 distinct bank pairs) is 2-way instead of 4-way
    // conflicted (SQ counters, r5: 45 % of this kernel's LDS cycles were conflict cycles), and every reader follows with a per-lane constant: the chain's
    // ds_read_b128 takes the other 16-byte half on the swizzled rows (still conflict-free), a transpose read permutes its four chunks per lane group.
#if PTR_B6_SWZ
    const uint32_t swz = (uint32_t)((j >> 2) & 1);
    const uint32_t wr_z = (uint32_t)(j * kB6ZRS) + 8u * ((uint32_t)g ^ (2u * swz));        // + 16 dt rows, + 32 tile bytes: this lane's 8 bytes of a 112-column image row
    const uint32_t wr_x = (uint32_t)(j * kB6XRS) + 8u * ((uint32_t)g ^ (2u * swz));
    const uint32_t rd_b = (uint32_t)(j * kB6ZRS + 32 * (g >> 1)) + 16u * (((uint32_t)g & 1u) ^ swz);       // chain B fragment: document j, 8 features from 32 s + 8 g
    const uint32_t tr_z = (uint32_t)((4 * g + (j >> 2)) * kB6ZRS) + 8u * (((uint32_t)j & 3u) ^ (2u * ((uint32_t)g & 1u)));     // transpose-read chunk of a 112-column image (row 4 g + (j >> 2): its swizzle bit is g & 1)
    const uint32_t tr_x = (uint32_t)((4 * g + (j >> 2)) * kB6XRS) + 8u * (((uint32_t)j & 3u) ^ (2u * ((uint32_t)g & 1u)));
#else
    const uint32_t wr_z = (uint32_t)(j * kB6ZRS + 8 * g);            // + 16 dt rows, + 32 tile bytes: this lane's 8 bytes of a 112-column image row
    const uint32_t wr_x = (uint32_t)(j * kB6XRS + 8 * g);
    const uint32_t rd_b = (uint32_t)(j * kB6ZRS + 16 * g);           // chain B fragment: document j, 8 features from 32 s + 8 g
    const uint32_t tr_z = (uint32_t)((4 * g + (j >> 2)) * kB6ZRS + 8 * (j & 3));     // transpose-read chunk of a 112-column image
    const uint32_t tr_x = (uint32_t)((4 * g + (j >> 2)) * kB6XRS + 8 * (j & 3));
#endif

    // the slab's fragment of tile t (16 features x 32 documents) of an image, k slot (G, e): e < 4 document 4 G + e, e >= 4 document 16 + 4 G + e - 4
    auto read_tr = [&](BFrag (&f)[3], uint32_t img_lane, int plane_bytes, int row_bytes, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_b *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t))));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_b *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t + 16 * row_bytes))));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            f[p].u[0] = l2[0]; f[p].u[1] = l2[1]; f[p].u[2] = h2[0]; f[p].u[3] = h2[1];
        }
    };
#ifdef PTR_B6_ABL_NOLDS      // timing-only ablation (wrong results): every LDS-sourced MFMA operand is ONE fragment read at kernel start — the operand reads are dead code
    BFrag ablf[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { ablf[p].q = *reinterpret_cast<lds_u32x4_b *>((uintptr_t)(lds0 + (uint32_t)(lane * 16 + p * 1024))); asm volatile("" : "+v"(ablf[p].q)); }
#define B6_LDSOP(x) ablf
#else
#define B6_LDSOP(x) x
#endif
    auto mma6 = [&](f32x4 &c, const BFrag (&af_)[3], const BFrag (&bf_)[3]) __attribute__((always_inline)) {
        const BFrag (&af)[3] = B6_LDSOP(af_); const BFrag (&bf)[3] = B6_LDSOP(bf_);
        c = B6_MFMA(af[0], bf[2], c); c = B6_MFMA(af[1], bf[1], c); c = B6_MFMA(af[2], bf[0], c);
        c = B6_MFMA(af[0], bf[1], c); c = B6_MFMA(af[1], bf[0], c); c = B6_MFMA(af[0], bf[0], c);
    };
    auto mma6w = [&](f32x4 &c, int w0, const BFrag (&bf_)[3]) __attribute__((always_inline)) {       // A = the W^T fragment kept in st[w0 .. w0 + 2]
        const BFrag (&bf)[3] = B6_LDSOP(bf_);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, st[w0]), a1 = __builtin_bit_cast(bf16x8, st[w0 + 1]), a2 = __builtin_bit_cast(bf16x8, st[w0 + 2]);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf[2].v, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[1].v, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bf[0].v, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf[1].v, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[0].v, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bf[0].v, c, 0, 0, 0);
    };
    // the 16-deep tail of chain layer cl: A = its two dwords per plane out of st[36..38], B = 8 bytes per plane (features 96 + 4 g .. + 3)
    auto mma6t = [&](f32x4 &c, int cl, const u32x2 (&bt)[3]) __attribute__((always_inline)) {       // cl: constant after unrolling
        i16x4 at[3], bb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int d = 2 * (3 * cl + p);
            at[p] = __builtin_bit_cast(i16x4, u32x2{__float_as_uint(st[36 + d / 4][d % 4]), __float_as_uint(st[36 + d / 4][d % 4 + 1])});
            bb[p] = __builtin_bit_cast(i16x4, bt[p]);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[0], bb[2], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[1], bb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[2], bb[0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[0], bb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[1], bb[0], c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[0], bb[0], c, 0, 0, 0);
    };
    // one dW row: A fragment = dZ tile `mo` of image zimg, B fragments = in-tiles n0 .. n0 + NN - 1 of image aimg.  PIPE: the next B fragment is
    // read while the current one is multiplied (12 registers more: only where the phase has them — not in chain 2, where X is in flight)
    auto dw_row = [&](auto nn_, auto pipe_, int acc0, uint32_t zimg, int mo, uint32_t aimg_lane, int a_plane, int a_row, int n0) __attribute__((always_inline)) {
        constexpr int NN = decltype(nn_)::value;
        constexpr bool PIPE = decltype(pipe_)::value;
        BFrag za[3];
        read_tr(za, zimg + tr_z + (uint32_t)(32 * mo), kB6ZPL, kB6ZRS, 0);
        if constexpr (PIPE && NN > 1) {
            BFrag ab[2][3];
            read_tr(ab[0], aimg_lane, a_plane, a_row, n0);
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                if (n + 1 < NN) read_tr(ab[(n + 1) & 1], aimg_lane, a_plane, a_row, n0 + n + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma6(st[acc0 + n], za, ab[n & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                BFrag ab[3];
                read_tr(ab, aimg_lane, a_plane, a_row, n0 + n);
                mma6(st[acc0 + n], za, ab);
                __builtin_amdgcn_sched_barrier(0);        // one fragment in flight: the scheduler otherwise issues every read of the row up front and spills
            }
        }
    };

    // ---- prefetch state: the stored activations of the next slab by DMA, dLoss/dscore in registers; X of the current slab in registers
    f32x4 xr[2][2];                        // X of the CURRENT slab, [tile slot][doc tile]: chain wave: slot 0 = tile w; wave 7: tiles 7, 8 — loaded
                                           // behind B1, turned into the XI image in front of B3 (it is only read by the dW_1 phase)
    float dsv[2];
    const b6_rsrc xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, TAIL ? 0 : (int)((uint32_t)R * (uint32_t)(F * 4)), 0x00020000);
    const b6_rsrc zsrd = __builtin_amdgcn_make_buffer_rsrc(dz0, 0, TAIL ? (int)((uint32_t)R * (uint32_t)(kAL * 4)) : 0, 0x00020000);      // host: R * 448 < 4 GB
    const b6_rsrc dsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dpreds), 0, (int)((uint32_t)R * 4u), 0x00020000);
    const b6_srd asrd = b6_make_srd(acts, (uint32_t)NL * (uint32_t)(act_layer_floats(R) * 4));       // host: NL * ceil16(R) * 448 and R * F * 4 < 4 GB
    uint32_t jx = (uint32_t)j * (uint32_t)(F * 4) + (uint32_t)g * 16, j4 = (uint32_t)j * 4, l16 = (uint32_t)lane * 16;
    asm volatile("" : "+v"(jx), "+v"(j4), "+v"(l16));
    auto load_x = [&](int slab) __attribute__((always_inline)) {
        const int row0 = slab * kB6S;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int t = chain ? W : 7 + sl;                    // rows past R: zeros; columns past F: the next row's numbers (selected away)
                if (!(chain && sl == 1))
                    xr[sl][dt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, (int)jx, (int)((uint32_t)(row0 + 16 * dt) * (uint32_t)(F * 4) + (uint32_t)(64 * t)), 0));
            }
    };
    // dropout key of the X site: row * kDropRowMul + fg * kDropFgMul + seed_lo with fg = 4 t + g — the lane part in ONE register, the rest scalar
    uint32_t jk = (uint32_t)j * kDropRowMul + (uint32_t)g * kDropFgMul + a.seed_lo;
    asm volatile("" : "+v"(jk));
    auto stage_x_tile = [&](int row0, int t_, const f32x4 (&x2)[2]) __attribute__((always_inline)) {
        const int t = (int)b6_opaque((uint32_t)t_);
        int col = 16 * t + 4 * g;
        asm volatile("" : "+v"(col));          // opaque: the column selects below are computed here, not hoisted as 24 lane masks (spilled SGPR pairs)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            uint32_t w0, w1;
            drop_bits_key(jk + ((uint32_t)(row0 + 16 * dt) * kDropRowMul + (uint32_t)(4 * t) * kDropFgMul), a.seed_hi, w0, w1);
            f32x4 v = drop4(x2[dt], w0, w1, thr, scale);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = col + r < F ? v[r] : (col + r == F ? 1.0f : 0.0f);      // the ones column: column F of dW_1 is db_1
            b6_write4(b6_opaque(lds0 + kB6_XI + (uint32_t)(32 * t)) + wr_x + (uint32_t)(16 * dt * kB6XRS), kB6XPL, v);
        }
    };
    auto stage_x = [&](int slab) __attribute__((always_inline)) {
        if (chain) stage_x_tile(slab * kB6S, W, xr[0]);
        else { stage_x_tile(slab * kB6S, 7, xr[0]); stage_x_tile(slab * kB6S, 8, xr[1]); }
    };
    // the next slab's stored activations by DMA into the staging area, its dLoss/dscore into registers
    auto prefetch = [&](int slab) __attribute__((always_inline)) {
        const int row0 = slab * kB6S;
        // a slab image is contiguous in a layer (two tile-major row tiles of 7 KB): 14 pieces of 1 KB per layer, 42 in all (a row tile past the
        // end of a layer reads the next layer's first tile / zeros: those documents carry dLoss/dscore = 0)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#if PTR_B6_DMA_CHAIN_ONLY
            const int k = chain ? W + 7 * q : 3 * 14;                // scalar; wave 7 (the longest MFMA stream of the chain phases) issues none
#else
            const int k = W + 8 * q;                                 // scalar
#endif
            if (k < 3 * 14) {
                const int layer = k / 14, ch = k - 14 * layer;
                b6_bdma16(asrd, l16, (uint32_t)layer * (uint32_t)(act_layer_floats(R) * 4) + (uint32_t)row0 * (kAL * 4) + (uint32_t)ch * 1024,
                          __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(kB6_ST + layer * kB6STG + ch * 1024)));
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)                               // rows past R: out of range, 0 — exactly the gradient they must contribute
            dsv[dt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsrd, (int)j4, (int)((uint32_t)(row0 + 16 * dt) * 4u), 0));
    };
    // r5: the slab loop is software-pipelined.  The staging pass of slab s + 1 (fp32 staging area -> dZ3 and the plane images of A2 / A1)
    // runs INSIDE the dW_1 phase of slab s: it writes the dZ buffer, A2 and A1, which the dW_1 phase (dZ1 x X) does not read — the two dZ
    // buffers swap roles every slab (`zi` = the buffer holding this slab's dZ3, then dZ1; `zo` = dZ2, then the NEXT slab's dZ3).  Waves w and
    // w + 4 share a SIMD: waves 0..3 stage first and multiply second, waves 4..7 the other way round, so a SIMD's matrix pipe and vector
    // ALU are busy at the same time.  Three barriers per slab (was four, with an all-VALU staging phase of 3.8 K cycles of 21.4 K).
    uint32_t m2 = 0u, m1 = 0u;                         // gate bits (activation > 0) of this lane's 2 x 4 elements of layers 2 / 1, per slab
    auto staging = [&](uint32_t zdst, int dt0 = 0, int dt1 = 2) __attribute__((always_inline)) {        // zdst: LDS offset (from lds0) of the dZ buffer that receives dZ3; document tiles dt0 .. dt1 - 1
        if (dt0 == 0) { m2 = 0u; m1 = 0u; }
#if defined(PTR_B6_ABL_NOSTAGE) || defined(PTR_B6_ABL_ST_NOMASK)     // (opaque gate bits: with a known 0 the compiler deletes the chain behind them)
        m2 = 0xa5u; m1 = 0x5au;
        asm volatile("" : "+v"(m2), "+v"(m1));
#endif
#ifdef PTR_B6_ABL_NOSTAGE    // timing-only ablation (wrong results): no staging pass at all
        if (false) {
#else
        if (chain) {
#endif
            const f32x4 wo4 = *reinterpret_cast<lds_f32x4_b *>((uintptr_t)(b6_opaque(lds0 + (uint32_t)(kB6_WO + 64 * W)) + (uint32_t)(16 * g)));
#ifndef PTR_B6_STAGE_PRELOAD
#define PTR_B6_STAGE_PRELOAD 0        /* experiment: both document tiles' staging-area reads issued up front (r5 / default: tile by tile — the second tile's reads wait behind the first tile's stores) */
#endif
#if PTR_B6_STAGE_PRELOAD
            f32x4 pre[2][3];
            {
                const uint32_t so0 = b6_opaque(lds0 + (uint32_t)(kB6_ST + 1024 * W)) + (uint32_t)(j * 64 + 16 * g);
#pragma unroll
                for (int dt = dt0; dt < dt1; ++dt)
#pragma unroll
                    for (int l = 0; l < 3; ++l) pre[dt][l] = *reinterpret_cast<lds_f32x4_b *>((uintptr_t)(so0 + (uint32_t)(dt * kActTile * 4 + l * kB6STG)));
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
            for (int dt = dt0; dt < dt1; ++dt) {
                // the staging area is a straight copy of two tile-major row tiles (ptr_mlp.h): feature tile W of row tile dt at dt * 7168 + W * 1024, lane (j, g) at j * 64 + 16 g
                const uint32_t so = b6_opaque(lds0 + (uint32_t)(kB6_ST + 1024 * W)) + (uint32_t)(j * 64 + 16 * g) + (uint32_t)(dt * kActTile * 4);
#ifdef PTR_B6_ABL_ST_NOREAD     // timing-only ablations of the staging pass (wrong results): no staging-area reads / no mask bits / no A1, A2 image stores
                f32x4 a1 = wo4, a2 = wo4, a3 = wo4;
                asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3));
#elif PTR_B6_STAGE_PRELOAD
                const f32x4 a1 = pre[dt][0], a2 = pre[dt][1], a3 = pre[dt][2];
                (void)so;
#else
                const f32x4 a1 = *reinterpret_cast<lds_f32x4_b *>((uintptr_t)so);
                const f32x4 a2 = *reinterpret_cast<lds_f32x4_b *>((uintptr_t)(so + kB6STG));
                const f32x4 a3 = *reinterpret_cast<lds_f32x4_b *>((uintptr_t)(so + 2 * kB6STG));
#endif
                const float ds = dsv[dt];
                f32x4 z3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z3[r] = a3[r] > 0.0f ? ds * wo4[r] : 0.0f;
                    awo[r] = fmaf(ds, a3[r], awo[r]);
#ifndef PTR_B6_ABL_ST_NOMASK
                    m2 |= (a2[r] > 0.0f ? 1u : 0u) << (4 * dt + r);
                    m1 |= (a1[r] > 0.0f ? 1u : 0u) << (4 * dt + r);
#endif
                }
                if (W == 0 && g == 0) abo += ds;
                const uint32_t wo = wr_z + b6_opaque(lds0 + (uint32_t)(32 * W)) + (uint32_t)(16 * dt * kB6ZRS);
                b6_write4(wo + b6_opaque(zdst), kB6ZPL, z3);
#ifndef PTR_B6_ABL_ST_NOWRITE
                b6_write4(wo + kB6_A2, kB6ZPL, a2);
                b6_write4(wo + kB6_A1, kB6ZPL, a1);
#endif
            }
        }
    };
    const int slab0 = blockIdx.x;
    prefetch(slab0 < nslabs ? slab0 : nslabs - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // zero fill + first staging visible to every wave
    B6_EDGE();                                                       // E3 first slab's activations landed (r6: gathering W^T BEHIND the first DMA instead of in front of it was measured — 12.5 K against 11.7 K cycles to this point: not kept)

#ifdef PTR_B6_TRACE       // experiment builds: shader-clock stamps of workgroup 0 behind its partial gradient (ws is sized for 2 partials per CU)
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(ws + (size_t)gridDim.x * np_stride) + W * 256;
    int nstamp = 0;
#define B6_STAMP() do { if (blockIdx.x == 0 && lane == 0 && nstamp < 256) trace[nstamp++] = clock64(); } while (0)
#else
#define B6_STAMP() do { } while (0)
#endif
#if defined(PTR_B6_TRACE) && defined(PTR_B6_TRACE2)     // finer stamps inside the phases (12 per slab instead of 7)
#define B6_STAMP2() B6_STAMP()
#else
#define B6_STAMP2() do { } while (0)
#endif
#if PTR_B6_PRIO
    if (W >= 4) __builtin_amdgcn_s_setprio(1);      // waves w and w + 4 share a SIMD and the older one wins every arbitration: static priority for the younger half
#endif
    uint32_t zi = kB6_ZA, zo = kB6_ZB;
    staging(zi);                                                     // slab 0
    b6_barrier();                                                    // images complete, staging area consumed
    {
        const int nxt = slab0 + (int)gridDim.x;
        prefetch(nxt < nslabs ? nxt : nslabs - 1);                  // (past the last slab: a redundant copy nobody stages — its dLoss/dscore is never used)
    }
    for (int slab = slab0; slab < nslabs; slab += gridDim.x) {
        B6_STAMP();
        // ---- chain 3 (dZ3 -> dZ2) + dW_3, chain 2 (dZ2 -> dZ1) + dW_2
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (!TAIL) { if (c == PTR_B6_X_PHASE) load_x(slab); }      // live through this phase only (registers), staged in front of its barrier
            const uint32_t zin = b6_opaque(lds0 + (c == 0 ? zi : zo)), zout = b6_opaque(lds0 + (c == 0 ? zo : zi) + (uint32_t)(32 * W));
            const uint32_t aim = b6_opaque(lds0 + (c == 0 ? kB6_A2 : kB6_A1));
            if (chain) {
              auto chain_part = [&]() __attribute__((always_inline)) {
                f32x4 cc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                auto read_b = [&](BFrag (&b)[3], int u) __attribute__((always_inline)) {       // u = 2 s + dt
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        b[p].q = *reinterpret_cast<lds_u32x4_b *>((uintptr_t)(zin + rd_b + (uint32_t)(p * kB6ZPL + 16 * (u & 1) * kB6ZRS + 64 * (u >> 1))));
                };
                u32x2 bt[2][3];                                   // the 16-deep tail's B operands (features 96 + 4 g .. + 3 of both document tiles)
                auto read_bt = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            bt[dt][p] = *reinterpret_cast<lds_u32x2_b *>((uintptr_t)(zin + wr_z + (uint32_t)(192 + p * kB6ZPL + 16 * dt * kB6ZRS)));
                };
                if (c != PTR_B6_X_PHASE || PTR_B6_PIPE_C2) {      // the chain phase without X in flight has the registers for a fragment in flight beside the one being multiplied
                    BFrag b[2][3];
                    read_b(b[0], 0);
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        if (u + 1 < 6) read_b(b[(u + 1) & 1], u + 1);
                        else if (PTR_B6_TAILPRE) read_bt();      // the last step has no fragment to prefetch: its 12 registers take the tail operands
                        __builtin_amdgcn_sched_barrier(0);
                        mma6w(cc[u & 1], 18 + 3 * (3 * c + (u >> 1)), b[u & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        BFrag b[3];
                        read_b(b, u);
                        if (PTR_B6_TAILPRE && u == 5) read_bt();
                        mma6w(cc[u & 1], 18 + 3 * (3 * c + (u >> 1)), b);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (!PTR_B6_TAILPRE) read_bt();
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    mma6t(cc[dt], c, bt[dt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                B6_STAMP2();                                      // chain MFMAs issued
                const uint32_t m = c == 0 ? m2 : m1;
#if PTR_B6_EPI_MIX
                if (!(TAIL && c == 1)) {
                    // the epilogue rides between the MFMAs of the dW row: tile n's six MFMAs carry chunk n — gating of document tile 0 | its split + image
                    // stores | gating of tile 1 | its split + stores | (nothing)
                    BFrag za[3], ab[2][3];
                    f32x4 dz = f32x4{0.f, 0.f, 0.f, 0.f};
                    const bool pipe = c != PTR_B6_X_PHASE;          // (compile-time after unrolling) the phase with X in flight has no registers for a fragment ahead
                    read_tr(za, zin + tr_z + (uint32_t)(32 * W), kB6ZPL, kB6ZRS, 0);
                    if (pipe) read_tr(ab[0], aim + tr_z, kB6ZPL, kB6ZRS, 0);
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        if (pipe) { if (n + 1 < 5) read_tr(ab[(n + 1) & 1], aim + tr_z, kB6ZPL, kB6ZRS, n + 1); }
                        else read_tr(ab[0], aim + tr_z, kB6ZPL, kB6ZRS, n);
                        __builtin_amdgcn_sched_barrier(0);
                        mma6(st[5 * c + n], za, ab[pipe ? (n & 1) : 0]);
                        if (n < 4) {
                            const int dt = n >> 1;
                            if ((n & 1) == 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) dz[r] = (m >> (4 * dt + r)) & 1u ? cc[dt][r] * scale : 0.0f;
                            } else {
                                b6_write4(zout + wr_z + (uint32_t)(16 * dt * kB6ZRS), kB6ZPL, dz);
                            }
#pragma unroll
                            for (int i_ = 0; i_ < 6; ++i_) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, PTR_B6_EPI_K, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    B6_STAMP2();
                    return;
                }
#endif
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    f32x4 dz;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz[r] = (m >> (4 * dt + r)) & 1u ? cc[dt][r] * scale : 0.0f;
                    if (TAIL && c == 1) {        // dZ_1 leaves the chip: row (slab, dt, j), features 16 W + 4 g .. + 3 (rows past R: out of range, dropped)
                        const uint32_t row = (uint32_t)(slab * kB6S + 16 * dt + j);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dz), zsrd, (int)(row < (uint32_t)R ? row * (kAL * 4) + (uint32_t)(64 * W + 16 * g) : 0xFFFFF000u), 0, 0);
                    } else {
                        b6_write4(zout + wr_z + (uint32_t)(16 * dt * kB6ZRS), kB6ZPL, dz);
                    }
                }
                B6_STAMP2();                                      // epilogue done (gating, split, image stores issued)
              };
              // dW of the layer whose dZ is the chain's INPUT: row w, in-tiles 0..4 of the activations below it
              auto dw_part = [&]() __attribute__((always_inline)) {
#if PTR_B6_EPI_MIX
                if (!(TAIL && c == 1)) return;                   // (done inside chain_part, interleaved with the epilogue)
#endif
                if (c == 0) dw_row(std::integral_constant<int, 5>{}, std::bool_constant<PTR_B6_X_PHASE != 0 || PTR_B6_PIPE_C2 != 0>{}, 0, zin, W, aim + tr_z, kB6ZPL, kB6ZRS, 0);
                else dw_row(std::integral_constant<int, 5>{}, std::bool_constant<PTR_B6_X_PHASE != 1 || PTR_B6_PIPE_C2 != 0>{}, 5, zin, W, aim + tr_z, kB6ZPL, kB6ZRS, 0);
              };
#if PTR_B6_CHAIN_ORDER
              // r6 experiment: the two waves of a SIMD (w, w + 4) run the phase's two independent halves in OPPOSITE order, so that one wave's epilogue
              // (vector ALU + LDS stores) runs beside the other's matrix instructions instead of beside its epilogue.  One copy of each body: a two-trip loop
              const bool chain_first = W < 4;
#pragma unroll 1
              for (int h = 0; h < 2; ++h) {
                  if ((h == 0) == chain_first) chain_part(); else dw_part();
                  __builtin_amdgcn_sched_barrier(0);
              }
#else
              chain_part();
              dw_part();
#endif
            } else {
                B6_STAMP2(); B6_STAMP2();
#if PTR_B6_W7HOLD
                // wave 7: in-tiles 5, 6 of every row.  r6: the two activation fragments are the SAME for all seven rows — read once per phase
                // (r5: seven times: 126 transpose reads per phase, now 54), only the dZ tile of the row streams, one row ahead of its MFMAs
                constexpr int ZD = PTR_B6_W7_DEPTH;
                BFrag ab5[3], ab6[3], zr[ZD][3];
                read_tr(zr[0], zin + tr_z, kB6ZPL, kB6ZRS, 0);
                read_tr(ab5, aim + tr_z, kB6ZPL, kB6ZRS, 5);
                read_tr(ab6, aim + tr_z, kB6ZPL, kB6ZRS, 6);
                if (ZD == 3) read_tr(zr[1], zin + tr_z + 32u, kB6ZPL, kB6ZRS, 0);
#pragma unroll
                for (int mo = 0; mo < 7; ++mo) {
                    if (mo + ZD - 1 < 7) read_tr(zr[(mo + ZD - 1) % ZD], zin + tr_z + (uint32_t)(32 * (mo + ZD - 1)), kB6ZPL, kB6ZRS, 0);
                    __builtin_amdgcn_sched_barrier(0);
#if PTR_B6_W7_ILV
                    {
                        f32x4 &c0 = st[14 * c + 2 * mo], &c1 = st[14 * c + 2 * mo + 1];
                        const BFrag (&af)[3] = zr[mo % ZD];
                        c0 = B6_MFMA(af[0], ab5[2], c0); c1 = B6_MFMA(af[0], ab6[2], c1); c0 = B6_MFMA(af[1], ab5[1], c0); c1 = B6_MFMA(af[1], ab6[1], c1);
                        c0 = B6_MFMA(af[2], ab5[0], c0); c1 = B6_MFMA(af[2], ab6[0], c1); c0 = B6_MFMA(af[0], ab5[1], c0); c1 = B6_MFMA(af[0], ab6[1], c1);
                        c0 = B6_MFMA(af[1], ab5[0], c0); c1 = B6_MFMA(af[1], ab6[0], c1); c0 = B6_MFMA(af[0], ab5[0], c0); c1 = B6_MFMA(af[0], ab6[0], c1);
                    }
#else
                    mma6(st[14 * c + 2 * mo], zr[mo % ZD], ab5);
                    mma6(st[14 * c + 2 * mo + 1], zr[mo % ZD], ab6);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
#pragma unroll
                for (int mo = 0; mo < 7; ++mo)       // wave 7: in-tiles 5, 6 of every row
                    dw_row(std::integral_constant<int, 2>{}, std::bool_constant<PTR_B6_PIPE_W7 != 0>{}, 14 * c + 2 * mo, zin, mo, aim + tr_z, kB6ZPL, kB6ZRS, 5);
#endif
            }
            if constexpr (!TAIL) { if (c == PTR_B6_X_PHASE) stage_x(slab); }     // the XI image (read by dW_1 only: free since the last B4)
            // (the DMA goes BEHIND the X staging: hipcc's waitcnt for the X registers does not count the asm DMA, a wait placed behind it would
            // wait for the whole prefetch)
#if PTR_B6_PREFETCH_LATE
            if (c == 0 && slab != slab0) {                           // the staging area was consumed before the last B4: the NEXT slab's activations
                const int nxt = slab + (int)gridDim.x;               // (the prologue issued slab0's successor itself)
                prefetch(nxt < nslabs ? nxt : nslabs - 1);
            }
#endif
            if (c == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the NEXT slab's staging area has landed (issued a phase ago)
            B6_STAMP();
            b6_barrier();                                            // B2 / B3
            B6_STAMP();
        }
        // ---- dW_1: dZ1 (in `zi`) x the X image  ||  the staging pass of the next slab (-> `zo`, A2, A1: nobody reads them in this phase)
        {
            const bool more = slab + (int)gridDim.x < nslabs;       // wave-uniform
            if constexpr (TAIL) {
                if (more) staging(zo);
            } else {
            const uint32_t za = b6_opaque(lds0 + zi), xi = b6_opaque(lds0 + kB6_XI) + tr_x;
            auto dw1 = [&]() __attribute__((always_inline)) {
                if (chain) {
                    dw_row(std::integral_constant<int, 8>{}, std::true_type{}, 10, za, W, xi, kB6XPL, kB6XRS, 0);
                } else {
#if PTR_B6_W7HOLD
                    BFrag xb[3], zr[2][3];                        // wave 7: in-tile 8 of X for every row — one fragment for the phase, the dZ tiles stream
                    read_tr(zr[0], za + tr_z, kB6ZPL, kB6ZRS, 0);
                    read_tr(xb, xi, kB6XPL, kB6XRS, 8);
#pragma unroll
                    for (int mo = 0; mo < 7; ++mo) {
                        if (mo + 1 < 7) read_tr(zr[(mo + 1) & 1], za + tr_z + (uint32_t)(32 * (mo + 1)), kB6ZPL, kB6ZRS, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        mma6(st[28 + mo], zr[mo & 1], xb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
#pragma unroll
                    for (int mo = 0; mo < 7; ++mo) dw_row(std::integral_constant<int, 1>{}, std::false_type{}, 28 + mo, za, mo, xi, kB6XPL, kB6XRS, 8);
#endif
                }
            };
#if PTR_B6_STAGE_ORDER == 4
            // complementary halves of a SIMD without a second copy of either body (two copies under a wave-id branch spilled 115 registers):
            // a two-trip loop that runs the staging pass in the first trip for waves 0..3 and in the second for waves 4..7 — waves w and
            // w + 4 share a SIMD, so its vector ALU splits one wave's next slab while its matrix pipe multiplies the other's dW_1
            const bool stage_first = W < 4;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                if ((h == 0) == stage_first) { if (more) staging(zo); }
                else dw1();
                __builtin_amdgcn_sched_barrier(0);
            }
#elif PTR_B6_STAGE_ORDER == 3
            // the two halves of the staging pass between the two halves of the dW_1 row: the MFMAs of four tiles drain while the vector ALU splits
            if (chain) {
                if (more) staging(zo, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                dw_row(std::integral_constant<int, 4>{}, std::true_type{}, 10, za, W, xi, kB6XPL, kB6XRS, 0);
                if (more) staging(zo, 1, 2);
                __builtin_amdgcn_sched_barrier(0);
                dw_row(std::integral_constant<int, 4>{}, std::true_type{}, 14, za, W, xi, kB6XPL, kB6XRS, 4);
            } else dw1();
#elif PTR_B6_STAGE_ORDER == 0
            if (more) staging(zo);
            __builtin_amdgcn_sched_barrier(0);
            B6_STAMP2();                                          // staging pass done
            dw1();
#elif PTR_B6_STAGE_ORDER == 1
            dw1();
            __builtin_amdgcn_sched_barrier(0);
            if (more) staging(zo);
#else
            if (W < 4) {
                if (more) staging(zo);
                __builtin_amdgcn_sched_barrier(0);
                dw1();
            } else {
                dw1();
                __builtin_amdgcn_sched_barrier(0);
                if (more) staging(zo);
            }
#endif
        }
            }
        B6_STAMP();
        b6_barrier();                                                // B4: dZ1 / X consumed, next slab's images complete, staging area free
        B6_STAMP();
#if !PTR_B6_PREFETCH_LATE
        {
            const int nxt = slab + 2 * (int)gridDim.x;
            prefetch(nxt < nslabs ? nxt : nslabs - 1);
        }
#endif
        const uint32_t tz = zi; zi = zo; zo = tz;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the run-ahead DMA must not outlive the workgroup's LDS allocation
    B6_EDGE();                                                       // E4 slab loop done

    // ---- this workgroup's partial gradient, flat parameter layout (every element written exactly once)
    float *out = ws + (size_t)blockIdx.x * np_stride;
    auto store_tile = [&](const f32x4 &c, int layer, int mo, int ni) __attribute__((always_inline)) {
        const int K = layer == 0 ? F : kH;
        const int in = 16 * ni + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * mo + 4 * g + r;
            if (o < kH) {
                if (in < K) out[off_W(layer, F) + (size_t)o * K + in] = c[r];
                else if (in == K) out[off_b(layer, F) + o] = c[r];
            }
        }
    };
    if (chain) {
#pragma unroll
        for (int n = 0; n < 5; ++n) { store_tile(st[n], 2, W, n); store_tile(st[5 + n], 1, W, n); }
        if constexpr (!TAIL) {
#pragma unroll
            for (int n = 0; n < 8; ++n) store_tile(st[10 + n], 0, W, n);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = awo[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
            const int f = 16 * W + 4 * g + r;
            if (j == 0 && f < kH) out[off_wout(NL, F) + f] = v;
        }
        if (W == 0) {
            float v = abo;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) out[off_wout(NL, F) + kH] = v;
        }
    } else {
#pragma unroll
        for (int mo = 0; mo < 7; ++mo) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { store_tile(st[2 * mo + q], 2, mo, 5 + q); store_tile(st[14 + 2 * mo + q], 1, mo, 5 + q); }
            if constexpr (!TAIL) store_tile(st[28 + mo], 0, mo, 8);
        }
    }
#ifdef PTR_B6_TRACE_EDGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B6_EDGE();                                                       // E5 partial gradient stored
#endif
}

// PTR_BWD_X6 (read per call): "0" selects the fp32-MFMA fused backward (scorer_bwd.hip) instead.  r5: this kernel is the DEFAULT wherever it
// serves the shape — software-pipelined (the next slab's staging pass inside the dW_1 phase, DMA issued by the chain waves at the end of the
// chain-3 phase) it takes 462 us at 524 288 x 136 against 587-603 us (r4: 610 vs 600, opt-in).
static int bwd_x6_mode() {
    const char *e = getenv("PTR_BWD_X6");
    return e ? atoi(e) : 1;
}
bool bwd_x6_supported(int R, int F, int NL, const void *X, const void *acts) {
    if (bwd_x6_mode() == 0) return false;
    const int NT1 = (F + 15) / 16;
    if ((uint64_t)NL * (uint64_t)act_layer_floats(R) * 4 >= 0xFFFFF000ull || (uint64_t)R * (uint64_t)F * 4 >= 0xFFFFF000ull) return false;     // buffer resources: < 4 GB
    return NL == 3 && NT1 == 9 && F % 4 == 0 && F < 16 * NT1 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(acts) & 15) == 0;
}
int launch_bwd_x6(const float *X, const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, hipStream_t st,
                  const char *who) {
    const int grid = bwd_fused_grid(a.R);
    const size_t NP = n_params(a.NL, a.F);
    auto kern = mlp_bwd_x6_kernel<9>;
    if (int e = allow_lds(kern, kB6Lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kB6Lds, st, X, params, acts, dpreds, a, ws, NP, (float *)nullptr);
    return check_hip(hipGetLastError(), who);
}

// The TAIL form (every input width, three hidden layers): dZ chain + the hidden layers' gradients + d w_out / d b_out in one pass over the stored
// activations on the bf16 instructions, dZ of the first layer written to dz0 [R][112] for the first-layer dW kernel.  Same grid and partial layout as
// the fp32-MFMA tail kernel (launch_bwd_tail, scorer_bwd.hip), which PTR_BWD_X6=0 keeps.
bool bwd_x6_tail_supported(int R, int NL, const void *acts) {
    if (bwd_x6_mode() == 0) return false;
    if ((uint64_t)NL * (uint64_t)act_layer_floats(R) * 4 >= 0xFFFFF000ull || (uint64_t)R * (kAL * 4) >= 0xFFFFF000ull) return false;      // buffer resources: < 4 GB
    return NL == 3 && (reinterpret_cast<uintptr_t>(acts) & 15) == 0;
}
int launch_bwd_x6_tail(const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, float *dz0, hipStream_t st,
                       const char *who) {
    const int grid = bwd_fused_grid(a.R);
    const size_t NP = n_params(a.NL, a.F);
    auto kern = mlp_bwd_x6_kernel<0>;
    if (int e = allow_lds(kern, kB6Lds)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kB6Lds, st, (const float *)nullptr, params, acts, dpreds, a, ws, NP, dz0);
    return check_hip(hipGetLastError(), who);
}

}  // namespace ptr
