// Fused Adam step for the hash tables / vectors / MLP weights (humanrf/run.py:101: torch.optim.Adam,
// betas (0.9, 0.99), eps 1e-15) with the GradScaler semantics of humanrf/trainer.py:250-252 folded in:
// gradients are un-scaled on the fly, the whole step is skipped when flags[0] != 0 (found_inf), the
// gradient buffer is zeroed for the next step and the fp16 copy the kernels gather from is refreshed --
// one pass over HBM (32 B/param) instead of unscale + step + half-cast + zero_grad passes.
#include "hrf_common.h"

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float beta1, float beta2,
                                         float eps, float bc2_sqrt, float inv_scale)
{
    const float gr = g * inv_scale;
    // torch/optim/adam.py (_single_tensor_adam): exp_avg.lerp_(grad, 1-beta1);
    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    m = m + (gr - m) * (1.0f - beta1);
    v = v * beta2 + (1.0f - beta2) * gr * gr;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// kVec: four parameters per thread through 16-byte accesses (all pointers 16-byte aligned, n % 4 handled by the
// scalar kernel on the tail). The fp32 streams are read and written once per step and are far larger than any cache,
// so they go through non-temporal accesses; the fp16 copy is what the gather kernels read next and stays cacheable.
template <bool kVec>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, __half* __restrict__ p16, int64_t n,
                                              float step_size, float beta1, float beta2, float eps, float bc2_sqrt,
                                              float inv_scale, const int32_t* __restrict__ flags)
{
    const bool skip = flags && flags[0] != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (kVec) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            if (!skip) {
                const f4v gi = __builtin_nontemporal_load((const f4v*)g + i);
                f4v pi = __builtin_nontemporal_load((const f4v*)p + i);
                f4v mi = __builtin_nontemporal_load((const f4v*)m + i);
                f4v vi = __builtin_nontemporal_load((const f4v*)v + i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float pk = pi[k], mk = mi[k], vk = vi[k];
                    adam_one(pk, gi[k], mk, vk, step_size, beta1, beta2, eps, bc2_sqrt, inv_scale);
                    pi[k] = pk; mi[k] = mk; vi[k] = vk;
                }
                __builtin_nontemporal_store(mi, (f4v*)m + i);
                __builtin_nontemporal_store(vi, (f4v*)v + i);
                __builtin_nontemporal_store(pi, (f4v*)p + i);
                if (p16) {
                    const __half2 lo = __floats2half2_rn(pi[0], pi[1]), hi = __floats2half2_rn(pi[2], pi[3]);
                    ((uint2*)p16)[i] = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
                }
            }
            __builtin_nontemporal_store(f4v{0.0f, 0.0f, 0.0f, 0.0f}, (f4v*)g + i);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            if (!skip) {
                float pi = p[i], mi = m[i], vi = v[i];
                adam_one(pi, g[i], mi, vi, step_size, beta1, beta2, eps, bc2_sqrt, inv_scale);
                m[i] = mi; v[i] = vi; p[i] = pi;
                if (p16) p16[i] = __float2half(pi);
            }
            g[i] = 0.0f;
        }
    }
}

extern "C" int hrf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* p16, int64_t n,
                             float lr, float beta1, float beta2, float eps, float bc1, float bc2, float grad_scale,
                             const int32_t* flags, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "NULL argument");
    HRF_CHECK_ARG(bc1 > 0.0f && bc2 > 0.0f && grad_scale > 0.0f, "bad bias corrections / scale");
    const float step_size = lr / bc1, bc2s = sqrtf(bc2), inv = 1.0f / grad_scale;
    const uintptr_t align = (uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq |
                            ((uintptr_t)p16 << 1);  // p16 needs 8-byte alignment for four halves
    int64_t done = 0;
    if ((align & 15u) == 0 && n >= 4) {
        const int64_t n4 = n >> 2;
        unsigned blocks = hrf_blocks(n4, 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_adam<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                           (__half*)p16, n, step_size, beta1, beta2, eps, bc2s, inv, flags);
        done = n4 << 2;
    }
    if (done < n) {  // unaligned buffers, or the last n % 4 parameters
        const int64_t rest = n - done;
        unsigned blocks = hrf_blocks(rest, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_adam<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param + done, grad + done,
                           exp_avg + done, exp_avg_sq + done, p16 ? (__half*)p16 + done : nullptr, rest, step_size, beta1,
                           beta2, eps, bc2s, inv, flags);
    }
    HRF_CHECK_LAUNCH();
    return 0;
}
