// Fused Adam step for the hash tables / vectors / MLP weights (humanrf/run.py:101: torch.optim.Adam,
// betas (0.9, 0.99), eps 1e-15) with the GradScaler semantics of humanrf/trainer.py:250-252 folded in:
// gradients are un-scaled on the fly, the whole step is skipped when flags[0] != 0 (found_inf), the
// gradient buffer is zeroed for the next step and the fp16 copy the kernels gather from is refreshed --
// one pass over HBM (32 B/param) instead of unscale + step + half-cast + zero_grad passes.
#include "hrf_common.h"
#include <algorithm>

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

// ------------------------------------------------------------------------------------------------
// All parameter tensors of the model in ONE launch, optimizer bookkeeping on the device, torch.optim.Adam's
// per-parameter semantics kept:
//   * the reference builds one Decomposition4D module per temporal segment and only runs the segments the batch touches
//     (humanrf.py:159-179); the others get no gradient (zero_grad(set_to_none=True), trainer.py:174), and Adam skips
//     parameters without a gradient: their moments do not decay, their value does not move, THEIR step count does not
//     advance. Tensors therefore belong to GROUPS (0 = always stepped: the two MLPs and the camera embeddings;
//     1 + s = tables and vectors of segment s), each with its own step count, stepped only when its `touched` flag is set
//     (by hrf_loss_fwd_bwd from the frames of the batch's rays; summed over ranks in data-parallel runs).
//   * GradScaler.step (trainer.py:250-252): with found_inf nothing moves and no step count advances.
// state (int32): [0] found_inf of this step (set by the backward kernels), [1] steps skipped, [2] workgroups finished
// (internal), [3] unused, [4 + g] steps taken by group g, [4 + G + g] touched flag of group g. The last workgroup to
// finish advances the counters and clears the flags for the next step: no host-side or extra-launch bookkeeping.
// Untouched tensors are not read at all (their gradients are already zero), so a step streams only the touched
// segments: 32 B per touched parameter instead of 32 B per parameter.
// ------------------------------------------------------------------------------------------------
#define ADAM_MAX_ACTIVE 256
// round to nearest even (v_cvt_pk_bf16_f32), same as mlp_common.h's hrf_f32_to_bf16
__device__ __forceinline__ short adam_bf16(float x) { return __builtin_bit_cast(short, (__bf16)x); }
// One entry per tensor that is stepped in this launch, written by k_adam_prepare (one thread walks the descriptors ONCE
// for the whole launch; a first version let thread 0 of each of the 8192 workgroups do this walk -- a chain of dependent
// global loads -- and spent 0.14 ms on it before a single parameter moved, micro-benchmark tools/adam_bench.py).
struct AdamActive {
    int64_t start4;      // first 16-byte chunk of this tensor in the launch's index space
    int64_t n;
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    void* p16;
    int32_t p16_bf16;    // the 16-bit copy holds bf16 (MLP weights of a bf16 model) instead of fp16
    int32_t pad_;
    float step_size;     // lr / (1 - beta1^t)
    float bc2_sqrt;      // sqrt(1 - beta2^t)
    int64_t bulk4;       // 16-byte chunks handled by the vector path (0 for unaligned tensors)
};
struct AdamPlan {
    int32_t active;
    int32_t skip;
    int64_t total4;
    float inv_scale;     // 1 / (grad_scale * scaler->scale) of THIS step
    float pad_[3];
    AdamActive t[ADAM_MAX_ACTIVE];
};

__global__ void k_adam_prepare(const hrf_adam_tensor* __restrict__ tensors, int count, int num_groups, float lr, float beta1,
                               float beta2, float grad_scale, const int32_t* __restrict__ state,
                               hrf_grad_scaler* __restrict__ scaler, AdamPlan* __restrict__ plan, int first_chunk)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool found_inf = state[0] != 0;
    float dyn = 1.0f;
    // (a step over more than ADAM_MAX_ACTIVE descriptors runs as several launches over slices of them: the scale is read
    // and updated by the first one, whose inv_scale stays in the plan; the bookkeeping is done by the last one)
    if (scaler && first_chunk) {
        // torch.amp.GradScaler: unscale with the scale the loss was multiplied by, then update() (trainer.py:251-252):
        // found_inf -> scale *= backoff_factor, tracker = 0; otherwise tracker += 1 and at growth_interval clean steps
        // scale *= growth_factor, tracker = 0 (torch/amp/grad_scaler.py, _amp_update_scale_). The next loss kernel on
        // this stream reads the updated value.
        dyn = scaler->scale;
        if (found_inf) { scaler->scale = dyn * scaler->backoff_factor; scaler->growth_tracker = 0; }
        else {
            const int32_t tr = scaler->growth_tracker + 1;
            if (tr >= scaler->growth_interval) { scaler->scale = dyn * scaler->growth_factor; scaler->growth_tracker = 0; }
            else scaler->growth_tracker = tr;
        }
    }
    if (first_chunk) plan->inv_scale = 1.0f / (grad_scale * dyn);
    const int32_t* steps = state + 4;
    const int32_t* touched = state + 4 + num_groups;
    const float l2b1 = log2f(beta1), l2b2 = log2f(beta2);
    int a = 0;
    int64_t total = 0;
    for (int k = 0; k < count && a < ADAM_MAX_ACTIVE; ++k) {
        const hrf_adam_tensor T = tensors[k];
        if (T.group != 0 && touched[T.group] == 0) continue;   // no gradient this step: Adam leaves the tensor alone
        const uintptr_t align = (uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq |
                                ((uintptr_t)T.p16 << 1);   // (NULL members of a zero-only entry do not disturb the test)
        const float tf = (float)(steps[T.group] + 1);
        // bias corrections 1 - beta^t (exp2 of t*log2(beta): ~1e-7 relative)
        const float bc1 = 1.0f - exp2f(tf * l2b1), bc2 = 1.0f - exp2f(tf * l2b2);
        AdamActive e;
        e.start4 = total; e.n = T.n; e.param = T.param; e.grad = T.grad; e.exp_avg = T.exp_avg; e.exp_avg_sq = T.exp_avg_sq;
        e.p16 = T.p16; e.p16_bf16 = T.reserved & 1; e.pad_ = 0; e.step_size = lr / bc1; e.bc2_sqrt = sqrtf(bc2);
        e.bulk4 = ((align & 15u) == 0) ? (T.n >> 2) : 0;
        total += e.bulk4;
        plan->t[a++] = e;
    }
    plan->active = a;
    plan->skip = found_inf;
    plan->total4 = total;
}

__global__ __launch_bounds__(256) void k_adam_multi(const AdamPlan* __restrict__ plan, int num_groups, float beta1, float beta2,
                                                    float eps, int32_t* __restrict__ state, int stride_mode, int last_chunk)
{
    __shared__ AdamActive s_t[ADAM_MAX_ACTIVE];
    const int active = plan->active;
    const float inv_scale = plan->inv_scale;
    const bool skip = plan->skip != 0;
    const int64_t total4 = plan->total4;
    {   // the plan's entries into LDS, all threads at once (one memory round trip)
        const int words = active * (int)(sizeof(AdamActive) / 4);
        const uint32_t* src = (const uint32_t*)plan->t;
        uint32_t* dst = (uint32_t*)s_t;
        for (int w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f4v __attribute__((ext_vector_type(4)));
    // Pointers that come out of memory are generic: typed as global so that the streams are global_load / global_store.
    typedef __attribute__((address_space(1))) f4v gf4v;
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned short gu16;
    // The 16-byte-aligned bulk of every stepped tensor forms ONE index space that the (small, persistent) grid strides
    // over -- or, with stride_mode 0, that is cut into one contiguous piece per workgroup: the work is spread evenly
    // whatever the mix of tensor sizes. A tensor's pointers are fetched from LDS when the index enters it.
    const int64_t piece = (total4 + gridDim.x - 1) / gridDim.x;
    const int64_t begin = stride_mode ? 0 : (int64_t)blockIdx.x * piece, end = stride_mode ? total4 : min(begin + piece, total4);
    const int64_t step = stride_mode ? stride : (int64_t)blockDim.x;
    int a = 0;
    while (a + 1 < active && begin >= s_t[a + 1].start4) ++a;
    int64_t idx = stride_mode ? tid : begin + threadIdx.x;
    while (idx < end) {
        while (a + 1 < active && idx >= s_t[a + 1].start4) ++a;
        const int64_t t_start = s_t[a].start4;
        const int64_t t_end = min(end, t_start + s_t[a].bulk4);
        gf4v* const gp = (gf4v*)s_t[a].grad;
        gf4v* const pp = (gf4v*)s_t[a].param;
        gf4v* const mp = (gf4v*)s_t[a].exp_avg;
        gf4v* const vp = (gf4v*)s_t[a].exp_avg_sq;
        gu64* const hp = (gu64*)s_t[a].p16;
        const bool hp_bf16 = s_t[a].p16_bf16 != 0;
        const float step_size = s_t[a].step_size, bc2_sqrt = s_t[a].bc2_sqrt;
        for (; idx < t_end; idx += step) {
            const int64_t i = idx - t_start;
            if (!skip && pp) {   // (pp == NULL: a gradient range another rank steps -- zeroed below, nothing else)
                const f4v gi = __builtin_nontemporal_load(gp + i);
                f4v pi = __builtin_nontemporal_load(pp + i);
                f4v mi = __builtin_nontemporal_load(mp + i);
                f4v vi = __builtin_nontemporal_load(vp + i);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float pk = pi[c], mk = mi[c], vk = vi[c];
                    adam_one(pk, gi[c], mk, vk, step_size, beta1, beta2, eps, bc2_sqrt, inv_scale);
                    pi[c] = pk; mi[c] = mk; vi[c] = vk;
                }
                __builtin_nontemporal_store(mi, mp + i);
                __builtin_nontemporal_store(vi, vp + i);
                __builtin_nontemporal_store(pi, pp + i);
                if (hp) {
                    uint32_t lo, hi;
                    if (hp_bf16) {
                        lo = (uint32_t)(uint16_t)adam_bf16(pi[0]) | ((uint32_t)(uint16_t)adam_bf16(pi[1]) << 16);
                        hi = (uint32_t)(uint16_t)adam_bf16(pi[2]) | ((uint32_t)(uint16_t)adam_bf16(pi[3]) << 16);
                    } else {
                        lo = __builtin_bit_cast(uint32_t, __floats2half2_rn(pi[0], pi[1]));
                        hi = __builtin_bit_cast(uint32_t, __floats2half2_rn(pi[2], pi[3]));
                    }
                    hp[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
                }
            }
            __builtin_nontemporal_store(f4v{0.0f, 0.0f, 0.0f, 0.0f}, gp + i);
        }
    }
    // what the bulk does not cover: unaligned tensors, and the last n % 4 parameters of every tensor
    for (int b = 0; b < active; ++b) {
        const int64_t done = s_t[b].bulk4 << 2, n = s_t[b].n;
        if (done >= n) continue;
        gfloat* p = (gfloat*)s_t[b].param;
        gfloat* g = (gfloat*)s_t[b].grad;
        gfloat* m = (gfloat*)s_t[b].exp_avg;
        gfloat* v = (gfloat*)s_t[b].exp_avg_sq;
        gu16* p16 = (gu16*)s_t[b].p16;
        for (int64_t i = done + tid; i < n; i += stride) {
            if (!skip && p) {
                float pi = p[i], mi = m[i], vi = v[i];
                adam_one(pi, g[i], mi, vi, s_t[b].step_size, beta1, beta2, eps, s_t[b].bc2_sqrt, inv_scale);
                m[i] = mi; v[i] = vi; p[i] = pi;
                if (p16) p16[i] = s_t[b].p16_bf16 ? (unsigned short)adam_bf16(pi) : __half_as_ushort(__float2half(pi));
            }
            g[i] = 0.0f;
        }
    }
    // bookkeeping by the last workgroup (every workgroup read the plan before it gets here)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&state[2], 1) == (int)gridDim.x - 1) {   // last workgroup of this launch
            if (last_chunk) {
                int32_t* st_steps = state + 4;
                int32_t* st_touched = state + 4 + num_groups;
                if (skip) state[1] += 1;
                for (int gidx = 0; gidx < num_groups; ++gidx) {
                    if (!skip && (gidx == 0 || st_touched[gidx] != 0)) st_steps[gidx] += 1;
                    if (gidx != 0) st_touched[gidx] = 0;
                }
                state[0] = 0;
            }
            state[2] = 0;
            __threadfence();
        }
    }
}

extern "C" size_t hrf_adam_workspace_bytes(void) { return sizeof(AdamPlan); }

extern "C" int hrf_adam_multi(const hrf_adam_tensor* tensors, int count, int num_groups, int64_t max_elements, float lr,
                              float beta1, float beta2, float eps, float grad_scale, int32_t* state,
                              hrf_grad_scaler* scaler, void* workspace, hrf_stream_t stream)
{
    HRF_CHECK_ARG(tensors && state && workspace, "NULL argument");
    HRF_CHECK_ARG(count > 0 && num_groups > 0 && max_elements >= 0, "bad counts");
    HRF_CHECK_ARG(grad_scale > 0.0f && beta1 > 0.0f && beta1 < 1.0f && beta2 > 0.0f && beta2 < 1.0f, "bad hyper-parameters");
    if (max_elements == 0) return 0;
    // Grid: ONE workgroup per CU striding over the index space of the touched tensors. Measured on MI355X
    // (tools/adam_bench.py, 39 M parameters, all segments touched): 256 workgroups 0.25 ms; 1024 0.32; 2048 0.34; 8192 0.72
    // -- every workgroup pays a fixed cost (plan fetch, release fence + ticket at the end); contiguous pieces per
    // workgroup instead of the grid stride: 0.30 at 256 workgroups.
    unsigned blocks = hrf_blocks((max_elements + 3) / 4, 256);
    if (blocks > 256u) blocks = 256u;
    // Any number of descriptors (a model of S temporal segments has 2 S + 3): slices of ADAM_MAX_ACTIVE per launch, the
    // scaler / step-count bookkeeping once (first / last slice). One launch up to 126 segments.
    for (int c0 = 0; c0 < count; c0 += ADAM_MAX_ACTIVE) {
        const int cnt = std::min(count - c0, ADAM_MAX_ACTIVE);
        const int first = c0 == 0, last = c0 + cnt >= count;
        hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, tensors + c0, cnt, num_groups, lr, beta1,
                           beta2, grad_scale, state, scaler, (AdamPlan*)workspace, first);
        hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const AdamPlan*)workspace,
                           num_groups, beta1, beta2, eps, state, 1, last);
    }
    HRF_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void k_uniform_fill(uint32_t seed, int64_t n, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hrf_uniform01(seed, (uint32_t)i);
}

extern "C" int hrf_uniform_fill(uint32_t seed, int64_t n, float* out, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(out && n < ((int64_t)1 << 32), "NULL output or more than 2^32 values");
    hipLaunchKernelGGL(k_uniform_fill, dim3(hrf_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, n, out);
    HRF_CHECK_LAUNCH();
    return 0;
}
