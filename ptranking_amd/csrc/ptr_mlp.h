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
int launch_dw_x6(const float *A, int lda, const float *dZ, int K, int nt_base, const MlpArgs &a, float *ws, size_t np_stride, size_t w_off, size_t b_off,
                 int nblk, hipStream_t st, const char *who);

}  // namespace ptr
