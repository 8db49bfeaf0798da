// Shared by linear.hip (fp32-MFMA linear layers) and linear_x6.hip (the same forward / backward-input product on the bf16 matrix instructions).
#pragma once
#include "ptr_device.h"

namespace ptr {

struct LinArgs {
    int R, K, N;
    int ldx, ldy, ldg;           // leading dimensions (floats) of X, Y and the gate
    int act;                     // PTR_LINEAR_* epilogue
    float p_drop;
    uint32_t seed_lo, seed_hi;
    int site;
};

// linear_x6.hip: Y = epi(X Wm^T + bias) with every fp32 product as six bf16 products (fp32 results).  Returns < 0 when the shape is not served
// (the caller then runs the fp32-MFMA kernel), 0 on success, a PTR_ERR_* code otherwise.  trans: Wm = W^T with W [K][N] (backward-input).
int launch_linear_x6(bool trans, const float *X, const float *W, const float *bias, const float *gate, const LinArgs &a, float *Y, int num_cus,
                     hipStream_t st, const char *who);

}  // namespace ptr
