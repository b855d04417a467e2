// nr_optim.hip -- the masked Adam update of the reference's optimizer: neural_renderer/optimizers.py:17-34.
// One elementwise kernel: an element whose gradient is exactly zero keeps its parameter AND its moments (vertices that no
// pixel saw in this step must not drift on stale momentum); otherwise the standard Adam step in float32, in the reference's
// operation order, with `lr` already holding alpha_t = alpha * sqrt(1 - beta2^t) / (1 - beta1^t) times the parameter's own
// learning-rate multiplier (optimizers.py:20).
#include "nr_device.h"

using namespace nr;

namespace {

__global__ __launch_bounds__(256) void k_adam_update(float *__restrict__ param, const float *__restrict__ grad,
                                                     float *__restrict__ m, float *__restrict__ v, size_t n, float lr,
                                                     float one_minus_beta1, float one_minus_beta2, float eps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    if (g != 0.0f) {  // optimizers.py:26 (a NaN gradient is "not zero" and propagates, as in the reference)
        float mi = m[i], vi = v[i];
        mi += one_minus_beta1 * (g - mi);      // :27
        vi += one_minus_beta2 * (g * g - vi);  // :28
        if (vi < 0.0f) vi = 0.0f;              // :29
        m[i] = mi;
        v[i] = vi;
        param[i] -= lr * mi / (sqrtf(vi) + eps);  // :30
    }
}

}  // namespace

NR_API int nr_adam_update(float *param, const float *grad, float *m, float *v, size_t count, float lr, float one_minus_beta1,
                          float one_minus_beta2, float eps, void *stream)
{
    if (!param || !grad || !m || !v) return NR_E_NULL;
    if (count == 0) return 0;
    if (count > (size_t)0x7fffffff * 256) return NR_E_SIZE;
    hipLaunchKernelGGL(k_adam_update, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, m,
                       v, count, lr, one_minus_beta1, one_minus_beta2, eps);
    return launch_status();
}
