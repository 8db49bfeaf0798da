// r6 scratch experiment: what the memory system gives a kernel shaped like linear_fwd_x6t_kernel — few long-lived waves (8 per CU), each alternating a burst of
// loads (a tile of X) and a burst of stores (a tile of Y) — against the access pattern of the burst (row-strided 64-byte pieces like the MFMA fragment layout,
// or one contiguous KB per instruction) and against the number of waves per CU.  No arithmetic: the loaded values are summed into the stored ones.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MODE 0: fragment pattern — lane (j = lane & 15, g = lane >> 4): row j of the tile, 16 bytes at column 4 g + 16 i (i-th instruction): 16 rows x 64 B per instruction
// MODE 1: contiguous — instruction i covers bytes [1024 i, 1024 i + 1024) of the tile
// MODE 2 / 3: pieces of 128 / 256 bytes per row: 8 / 4 rows per instruction (lane = (row r = lane / LPR, 16 bytes l = lane % LPR), LPR = 8 / 16 lanes per row)
template <int MODE, int NLD, int NST>
__global__ void __launch_bounds__(1024) stream_kernel(const float *__restrict__ X, float *__restrict__ Y, int ldx, int ldy, int ntiles, int rows_per_tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, j = lane & 15, g = lane >> 4;
    constexpr int LPR = MODE == 2 ? 8 : 16, RPI = 64 / LPR, CPI = 4 * LPR;      // lanes per row, rows and columns per instruction
    constexpr int NRG = 32 / RPI;                                               // row groups of a 32-row tile
    for (int tile = blockIdx.x * nw + wave; tile < ntiles; tile += gridDim.x * nw) {
        const size_t row0 = (size_t)tile * rows_per_tile;
        f32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if constexpr (MODE == 0) v[i] = *reinterpret_cast<const f32x4 *>(X + (row0 + (i & 1) * 16 + j) * ldx + 4 * g + 16 * (i >> 1));
            else if constexpr (MODE == 1) v[i] = *reinterpret_cast<const f32x4 *>(X + row0 * ldx + (size_t)256 * i + 4 * lane);
            else v[i] = *reinterpret_cast<const f32x4 *>(X + (row0 + (i % NRG) * RPI + lane / LPR) * ldx + CPI * (i / NRG) + 4 * (lane % LPR));
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NLD; ++i) s += v[i];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const f32x4 o = s + (float)i;
            if constexpr (MODE == 0) *reinterpret_cast<f32x4 *>(Y + (row0 + (i & 1) * 16 + j) * ldy + 4 * g + 16 * (i >> 1)) = o;
            else if constexpr (MODE == 1) *reinterpret_cast<f32x4 *>(Y + row0 * ldy + (size_t)256 * i + 4 * lane) = o;
            else *reinterpret_cast<f32x4 *>(Y + (row0 + (i % NRG) * RPI + lane / LPR) * ldy + CPI * (i / NRG) + 4 * (lane % LPR)) = o;
        }
    }
}
extern "C" int run_stream(int mode, int waves, int grid, const float *X, float *Y, int ldx, int ldy, int ntiles, int rows_per_tile, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    // 32 rows x 136 floats per tile = 17 408 B = 17 KB-instructions (16 B per lane): NLD = NST = 17 (mode 0 uses 18: nine 16-column groups x two 16-row halves; the
    // last group reads / writes 8 columns past 136 — the buffers are padded)
    if (mode == 0) stream_kernel<0, 18, 18><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);
    else if (mode == 1) stream_kernel<1, 17, 17><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);
    else if (mode == 2) stream_kernel<2, 20, 20><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);      // 4 row groups x 5 column groups of 32 (160 >= 136: padded buffers)
    else if (mode == 3) stream_kernel<3, 24, 24><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);        // 8 row groups x 3 column groups of 64 (192 columns: padded)
    // like for like on 128-column rows (512 B, every piece aligned, every byte touched): 16 instructions per 32 x 128 tile either way
    else if (mode == 4) stream_kernel<0, 16, 16><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);        // 16 rows x 64 B
    else if (mode == 5) stream_kernel<2, 16, 16><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);        // 8 rows x 128 B
    else stream_kernel<3, 16, 16><<<grid, waves * 64, 0, s>>>(X, Y, ldx, ldy, ntiles, rows_per_tile);                       // 4 rows x 256 B
    return (int)hipGetLastError();
}
