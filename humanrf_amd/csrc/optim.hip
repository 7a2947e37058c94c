// Fused Adam step for the hash tables / vectors / MLP weights (humanrf/run.py:101: torch.optim.Adam,
// betas (0.9, 0.99), eps 1e-15) with the GradScaler semantics of humanrf/trainer.py:250-252 folded in:
// gradients are un-scaled on the fly, the whole step is skipped when flags[0] != 0 (found_inf), the
// gradient buffer is zeroed for the next step and the fp16 copy the kernels gather from is refreshed --
// one pass over HBM (32 B/param) instead of unscale + step + half-cast + zero_grad passes.
#include "hrf_common.h"

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, __half* __restrict__ p16, int64_t n,
                                              float step_size, float beta1, float beta2, float eps, float bc2_sqrt,
                                              float inv_scale, const int32_t* __restrict__ flags)
{
    const bool skip = flags && flags[0] != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!skip) {
            const float gr = g[i] * inv_scale;
            // torch/optim/adam.py (_single_tensor_adam): exp_avg.lerp_(grad, 1-beta1);
            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
            const float mi = m[i] + (gr - m[i]) * (1.0f - beta1);
            const float vi = v[i] * beta2 + (1.0f - beta2) * gr * gr;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            const float pi = p[i] - step_size * (mi / denom);
            m[i] = mi; v[i] = vi; p[i] = pi;
            if (p16) p16[i] = __float2half(pi);
        }
        g[i] = 0.0f;
    }
}

extern "C" int hrf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* p16, int64_t n,
                             float lr, float beta1, float beta2, float eps, float bc1, float bc2, float grad_scale,
                             const int32_t* flags, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "NULL argument");
    HRF_CHECK_ARG(bc1 > 0.0f && bc2 > 0.0f && grad_scale > 0.0f, "bad bias corrections / scale");
    unsigned blocks = hrf_blocks(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (__half*)p16, n, lr / bc1, beta1, beta2, eps, sqrtf(bc2), 1.0f / grad_scale, flags);
    HRF_CHECK_LAUNCH();
    return 0;
}
