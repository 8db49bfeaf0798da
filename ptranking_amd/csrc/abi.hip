// C-ABI plumbing: thread-local error text, argument checks, the deterministic slot reduction.
#include <stdarg.h>
#include <stdio.h>

#include "ptr_device.h"

namespace ptr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return (int)e;
}

int check_batch(const void *preds, const void *second, int B, int L, const char *who) {
    if (B < 0 || L <= 0) { set_error("%s: bad shape B=%d L=%d", who, B, L); return PTR_ERR_INVALID_ARG; }
    if (B > 0 && (!preds || !second)) { set_error("%s: NULL input pointer", who); return PTR_ERR_INVALID_ARG; }
    if (L > PTR_MAX_LIST_LEN) {
        set_error("%s: list length %d exceeds PTR_MAX_LIST_LEN=%d", who, L, PTR_MAX_LIST_LEN);
        return PTR_ERR_UNSUPPORTED;
    }
    return 0;
}

// out[0] = scale * sum(x[0..n)) with a fixed tree (ptr_device.h block1024_sum): one workgroup of 1024 threads, sixteen loads in flight
// per thread.  n is a batch size (<= a few 10^5): ~4 us at 65 536 slots.
__global__ void __launch_bounds__(1024) sum_f32_kernel(const float *__restrict__ x, int n, float scale, float *__restrict__ out) {
    __shared__ float red[16];
    const float tot = block1024_sum(x, n, red);
    if (threadIdx.x == 0) out[0] = scale * tot;
}

}  // namespace ptr

extern "C" int ptr_abi_version(void) { return PTR_ABI_VERSION; }

extern "C" const char *ptr_last_error(void) { return ptr::g_err; }

extern "C" int ptr_sum_f32(const float *x, int n, float scale, float *out, void *stream) {
    if (n < 0 || !out || (n > 0 && !x)) { ptr::set_error("ptr_sum_f32: bad arguments"); return PTR_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(ptr::sum_f32_kernel, dim3(1), dim3(1024), 0, ptr::as_stream(stream), x, n, scale, out);
    return ptr::check_hip(hipGetLastError(), "ptr_sum_f32");
}
