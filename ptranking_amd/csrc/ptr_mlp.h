// Shared definitions of the fused pointsf scorer kernels (scorer.hip: forward, layer-wise backward, Adam; scorer_bwd.hip: the
// single-pass fused backward).  Reference: ptranking/base/point_ranker.py:30-55, ptranking/base/utils.py:288-356.
#pragma once
#include <stdlib.h>
#include <utility>

#include "ptr_device.h"
#include "ptr_dropout.h"

namespace ptr {

constexpr int kH = 100;          // hidden width, hard-wired in the reference (point_ranker.py:30)
constexpr int kHP = 112;         // padded to 7 MFMA tiles of 16
constexpr int kMT = 7;
constexpr int kMaxLayers = 8;
constexpr int kAL = 112;         // floats per row of the stored activations / leading dimension of the dZ scratch; features 100..111 are padding
// r5: the stored activations are TILE-MAJOR: [layer][row tile of 16][feature tile of 16 (7 of them)][row in tile][feature in tile] — every
// (16 rows x 16 features) block is ONE contiguous KB, the unit a forward store instruction writes (lane (j, g) = row j, features 4g..4g+3 of
// the tile: 64 lanes x 16 B back to back) and an LDS-DMA piece of the backward reads.  Row-major rows (448 B) made every store instruction
// touch sixteen 64-byte pieces in sixteen different rows (bf16x6 training forward 375 -> 349 us at 524 288 x 136, scratch/r5_call1.sh).
// A layer holds ceil(R / 16) whole row tiles: rows past R of the last tile are written (finite values) and never contribute (their
// dLoss/dscore is 0).  The dZ scratch of the layer-wise backward stays row-major [R][112].
constexpr int kActTile = 16 * kAL;                                   // floats of one row tile (7 KB)
__host__ __device__ inline int act_row_tiles(int R) { return (R + 15) >> 4; }
__host__ __device__ inline size_t act_layer_floats(int R) { return (size_t)act_row_tiles(R) * kActTile; }
__host__ __device__ inline size_t act_off(int row, int col) {       // float offset of (row, col) inside a layer
    return (size_t)(row >> 4) * kActTile + (size_t)(((col >> 4) << 8) + ((row & 15) << 4) + (col & 15));
}

// flat parameter layout:  W1[100][F] b1[100] | W2[100][100] b2[100] | ... | w_out[100] b_out[1]
__host__ __device__ inline size_t off_W(int l, int F) { return l == 0 ? 0 : (size_t)kH * F + kH + (size_t)(l - 1) * (kH * kH + kH); }
__host__ __device__ inline size_t off_b(int l, int F) { return off_W(l, F) + (l == 0 ? (size_t)kH * F : (size_t)kH * kH); }
__host__ __device__ inline size_t off_wout(int NL, int F) { return off_W(NL, F); }
__host__ __device__ inline size_t n_params(int NL, int F) { return off_wout(NL, F) + kH + 1; }

struct MlpArgs {
    int R, F, NL;
    float p_drop;            // 0 => no dropout (eval mode)
    uint32_t seed_lo, seed_hi;
};

int mlp_num_cus();

// ---- the bf16x6 forward's pre-split weight image (scorer_x6.hip): one slice = [3 planes][7 tiles][lane group g][16 out-features][8 k-slots] bf16
constexpr int kX6Rows = kHP;                       // 112 A rows per slice
constexpr int kX6PlaneBytes = kX6Rows * 32 * 2;    // 7168
constexpr int kX6SliceBytes = 3 * kX6PlaneBytes;   // 21504 = 21 DMA pieces of 1 KB
__host__ __device__ inline int x6_n1(int F) { return (F + 31) / 32; }
__host__ __device__ inline int x6_nslices(int F, int NL) { return x6_n1(F) + 4 * (NL - 1); }
#if defined(__HIPCC__)
// r6: ONE parameter's three bf16 planes written into the image at the place x6_prep_kernel puts them — the optimiser step of the fused train step
// (reduce_partials_kernel) refreshes the image element by element, so the next step's forward needs no prep launch.  Same split as scorer_x6.hip
// split_pack2 (round to nearest, v_cvt_pk_bf16_f32), element for element: the image is bit-identical to a fresh x6_prep_kernel run.
// i = index into the flat parameter vector; b_1, w_out and b_out are not part of the image (they live in LDS / registers of the forward).
__device__ __forceinline__ void x6_img_put(uint8_t *__restrict__ img, int F, int NL, size_t i, float x) {
    using bf2 = __attribute__((ext_vector_type(2))) __bf16;
    using f2 = __attribute__((ext_vector_type(2))) float;
    const size_t w1 = (size_t)kH * F;
    int sl, row, g, e;
    if (i < w1) {                                                    // W1[row][k]: slice k / 32, slot (g, e) <-> feature 32 s + 8 g + e
        row = (int)(i / F);
        const int k = (int)(i - (size_t)row * F), kk = k & 31;
        sl = k >> 5; g = kk >> 3; e = kk & 7;
    } else {
        if (i < w1 + kH) return;                                     // b_1
        const size_t jv = i - (w1 + kH);
        const int l = 1 + (int)(jv / (kH * kH + kH));
        if (l > NL - 1) return;                                      // w_out, b_out
        const int jj = (int)(jv - (size_t)(l - 1) * (kH * kH + kH));
        int k;
        if (jj < kH * kH) { row = jj / kH; k = jj - row * kH; } else { row = jj - kH * kH; k = kH; }      // the bias rides on the ones slot (feature 100)
        const int kk = k & 31;
        sl = x6_n1(F) + 4 * (l - 1) + (k >> 5);
        g = (kk & 15) >> 2; e = 4 * (kk >> 4) + (kk & 3);           // hidden layers: slot (g, e) <-> feature 32 s + 16 (e >> 2) + 4 g + (e & 3)
    }
    const uint16_t p1 = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x, 0.0f}, bf2)) & 0xffffu);
    const float r = x - __uint_as_float((uint32_t)p1 << 16);
    const uint16_t p2 = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(f2{r, 0.0f}, bf2)) & 0xffffu);
    const float t = r - __uint_as_float((uint32_t)p2 << 16);
    const uint16_t p3 = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(f2{t, 0.0f}, bf2)) & 0xffffu);
    uint8_t *at = img + (size_t)sl * kX6SliceBytes + (size_t)(row >> 4) * 1024 + g * 256 + (row & 15) * 16 + e * 2;
    *reinterpret_cast<uint16_t *>(at) = p1;
    *reinterpret_cast<uint16_t *>(at + kX6PlaneBytes) = p2;
    *reinterpret_cast<uint16_t *>(at + 2 * kX6PlaneBytes) = p3;
}
#endif

// ---- compile-time loop with a constexpr index: f(std::integral_constant<int, I>{}) for I = 0..N-1
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn &&f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// ---- single-pass fused backward (scorer_bwd.hip)
// true when the (F, NL) configuration and the pointer alignments are served by the fused kernel
bool bwd_fused_supported(int F, int NL, const void *X, const void *acts);
// enqueue; ws must hold bwd_fused_grid(R) * n_params floats; returns the number of per-block partials written (0 on error, error set)
int bwd_fused_grid(int R);
int launch_bwd_fused(const float *X, const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws,
                     hipStream_t st, const char *who);

// the TAIL form for wide inputs: everything but the first layer's weight gradient, dZ of the first layer to dz0 [R][112]; same grid / partials
bool bwd_tail_supported(int NL, const void *acts);
int launch_bwd_tail(const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, float *dz0, hipStream_t st,
                    const char *who);

// ---- the same on bf16 matrix instructions with fp32 results (scorer_bwd_x6.hip); same grid, same partial layout
bool bwd_x6_supported(int R, int F, int NL, const void *X, const void *acts);
int launch_bwd_x6(const float *X, const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, hipStream_t st,
                  const char *who);

bool bwd_x6_tail_supported(int R, int NL, const void *acts);
int launch_bwd_x6_tail(const float *params, const float *acts, const float *dpreds, const MlpArgs &a, float *ws, float *dz0, hipStream_t st,
                       const char *who);

// ---- bf16x6 row contraction dW of the first layer for wide inputs (scorer_dw_x6.hip); interface of mlp_bwd_dw_lds_kernel
bool dw_x6_supported(int R, int K, int lda, const void *A);
int launch_dw_x6(const float *A, int lda, const float *dZ, int K, int ntk, const MlpArgs &a, float *ws, size_t np_stride, size_t w_off, size_t b_off,
                 int nblk, hipStream_t st, const char *who);

// ---- internals the one-call train step (train_step.hip) chains: the extern "C" entry points with the image hand-over of r6
//   mlp_forward_x6_impl: ptr_mlp_forward_x6 with prep = false when `wimg` already holds the planes of `params`
//   mlp_backward_step_impl: ptr_mlp_backward_step whose optimiser step also refreshes `wimg` (NULL: plain ptr_mlp_backward_step)
int mlp_forward_x6_impl(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed, float *preds, float *acts,
                        void *wimg, bool prep, void *stream);
int mlp_backward_step_impl(const float *X, float *params, const float *acts, const float *dpreds, int R, int F, int NL, float p_drop, uint64_t seed, float *dz,
                           float *ws, float *grad, int opt_kind, float lr, float hyper1, float hyper2, float eps, float weight_decay, int step, float *state1,
                           float *state2, const float *loss_q, int nq, float *loss_out, void *wimg, void *stream);

}  // namespace ptr
