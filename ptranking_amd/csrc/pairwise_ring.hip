// lambdarank_ring_kernel<1 / 2 / 4>: the fused LambdaRank loss + gradient kernel for lists of up to 256 documents (the north-star kernel), in its
// own translation unit (per-source compiler flags can be tried on it alone: ptranking_amd/build.py EXTRA_FLAGS; see ptr_ring.h).
// Reference: ptranking/ltr_adhoc/listwise/lambdarank.py:39-56.
#include "ptr_ring.h"

namespace ptr {

int launch_lambdarank_ring_small(int dpt, int QPB, const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma, float *loss_q,
                                 float *grad, hipStream_t st) {
    const size_t lds = (size_t)QPB * 2 * 64 * dpt * sizeof(float);
    const dim3 grid((B + QPB - 1) / QPB), block(QPB * kWave);
    if (dpt == 1) hipLaunchKernelGGL(lambdarank_ring_kernel<1>, grid, block, lds, st, preds, labels, lens, B, L, sigma, loss_q, grad);
    else if (dpt == 2) hipLaunchKernelGGL(lambdarank_ring_kernel<2>, grid, block, lds, st, preds, labels, lens, B, L, sigma, loss_q, grad);
    else hipLaunchKernelGGL(lambdarank_ring_kernel<4>, grid, block, lds, st, preds, labels, lens, B, L, sigma, loss_q, grad);
    return (int)hipGetLastError();
}

}  // namespace ptr
