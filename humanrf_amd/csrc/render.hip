// Alpha-composited volume rendering for gfx950: replaces the nerfacc 0.3.1 calls of
// humanrf/volume_rendering.py:75-84 (render_visibility + boolean-mask compaction) and :123-145
// (render_weight_from_density, accumulate_along_rays x2, background blend), plus the loss of
// humanrf/trainer.py:205-247. Semantics: SURVEY.md A.4.
//
// ray_indices coming out of the sampler are sorted with one contiguous run per ray, so every kernel here
// is "one wavefront per ray": lanes take 64 consecutive samples of the run (coalesced loads), per-ray
// prefix quantities are wave scans, per-ray outputs are wave reductions -- no atomics, no index_add_.
#include "hrf_common.h"

__device__ __forceinline__ float wave_incl_scan(float v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ float wave_suffix_scan(float v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v += o;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ray_offsets(const int64_t* __restrict__ sample_ray, int64_t n, int64_t num_rays,
                                                     int32_t* __restrict__ ray_start)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_rays) return;
    int64_t lo = 0, hi = n;  // lower_bound(sample_ray, r)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sample_ray[mid] < r) lo = mid + 1; else hi = mid;
    }
    ray_start[r] = (int32_t)lo;
}

extern "C" int hrf_ray_offsets(const int64_t* sample_ray, int64_t n, int64_t num_rays, int32_t* out_ray_start,
                               hrf_stream_t stream)
{
    HRF_CHECK_ARG(out_ray_start && (n == 0 || sample_ray) && num_rays >= 0, "bad arguments");
    hipLaunchKernelGGL(k_ray_offsets, dim3(hrf_blocks(num_rays + 1, 256)), dim3(256), 0, (hipStream_t)stream, sample_ray,
                       n, num_rays, out_ray_start);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// render_visibility: T_i = prod_{j<i} (1 - alpha_j) as a SEQUENTIAL fp32 product in sample order (the
// order oracle/sampler_oracle.c:orc_visibility fixes), visible = (T >= eps) && (alpha >= thre).
// The 64 alphas of a chunk are loaded coalesced; the running product walks the lanes with readlane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_visibility(const float* __restrict__ alphas, const float* __restrict__ sigma,
                                                    const int32_t* __restrict__ ray_start, int64_t num_rays, float step,
                                                    float eps, float thre, uint8_t* __restrict__ out_vis,
                                                    int32_t* __restrict__ out_kept)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    float T = 1.0f;
    int32_t kept = 0;
    for (int32_t c0 = b; c0 < e; c0 += 64) {
        const int32_t i = c0 + lane;
        float a = 0.0f;
        if (i < e) a = alphas ? alphas[i] : (1.0f - expf(-sigma[i] * step));
        const float om = 1.0f - a;
        float myT = 0.0f;
        const int cnt = min(64, e - c0);
        for (int k = 0; k < cnt; ++k) {
            if (lane == k) myT = T;
            T = T * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, om), k));
        }
        const bool vis = (i < e) && (myT >= eps) && (a >= thre);
        if (i < e) out_vis[i] = vis ? 1 : 0;
        kept += __popcll(__ballot(vis));
        if (T < eps) {  // every later sample has T_i <= T < eps: invisible (prefix property)
            for (int32_t j = c0 + 64 + lane; j < e; j += 64) out_vis[j] = 0;
            break;
        }
    }
    if (lane == 0 && out_kept) out_kept[r] = kept;
}

extern "C" int hrf_visibility(const float* alphas, const float* sigma, const int32_t* ray_start, int64_t num_rays,
                              float step, float early_stop_eps, float alpha_thre, uint8_t* out_vis, int32_t* out_kept,
                              hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG((alphas || sigma) && ray_start && out_vis, "NULL argument");
    hipLaunchKernelGGL(k_visibility, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, alphas,
                       sigma, ray_start, num_rays, step, early_stop_eps, alpha_thre, out_vis, out_kept);
    HRF_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void k_compact_samples(const uint8_t* __restrict__ vis, const int32_t* __restrict__ slot,
                                                         const float* __restrict__ t, const int64_t* __restrict__ sample_ray,
                                                         int64_t n, float* __restrict__ out_t,
                                                         int64_t* __restrict__ out_ray)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !vis[i]) return;
    const int32_t s = slot[i];
    out_t[s] = t[i];
    out_ray[s] = sample_ray[i];
}

extern "C" int hrf_compact_samples(const uint8_t* vis, const int32_t* slot, const float* t, const int64_t* sample_ray,
                                   int64_t n, float* out_t, int64_t* out_sample_ray, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(vis && slot && t && sample_ray && out_t && out_sample_ray, "NULL argument");
    hipLaunchKernelGGL(k_compact_samples, dim3(hrf_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, vis, slot, t,
                       sample_ray, n, out_t, out_sample_ray);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// composite forward:  w_i = T_i (1 - exp(-sigma_i dt_i)),  T_i = exp(-sum_{j<i} sigma_j dt_j),
// dt_i = (t_i + step) - t_i in fp32 exactly as volume_rendering.py:124-125 feeds nerfacc.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_fwd(const float* __restrict__ sigma, const __half* __restrict__ rgb,
                                                       const float* __restrict__ t, const int32_t* __restrict__ ray_start,
                                                       const float* __restrict__ background, int64_t num_rays, float step,
                                                       float* __restrict__ out_color, float* __restrict__ out_acc)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    float carry = 0.0f, c0s = 0.0f, c1s = 0.0f, c2s = 0.0f, as = 0.0f;
    for (int32_t base = b; base < e; base += 64) {
        const int32_t i = base + lane;
        float sd = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (i < e) {
            const float ti = t[i];
            sd = sigma[i] * ((ti + step) - ti);
            cr = __half2float(rgb[i * 3 + 0]); cg = __half2float(rgb[i * 3 + 1]); cb = __half2float(rgb[i * 3 + 2]);
        }
        const float incl = wave_incl_scan(sd, lane);
        const float T = expf(-(carry + (incl - sd)));
        const float w = (i < e) ? T * (1.0f - expf(-sd)) : 0.0f;
        c0s += w * cr; c1s += w * cg; c2s += w * cb; as += w;
        carry += __shfl(incl, 63, 64);
    }
    c0s = wave_sum(c0s); c1s = wave_sum(c1s); c2s = wave_sum(c2s); as = wave_sum(as);
    if (lane == 0) {
        if (background) {  // volume_rendering.py:144-145
            const float om = 1.0f - as;
            c0s = c0s + background[r * 3 + 0] * om;
            c1s = c1s + background[r * 3 + 1] * om;
            c2s = c2s + background[r * 3 + 2] * om;
        }
        out_color[r * 3 + 0] = c0s; out_color[r * 3 + 1] = c1s; out_color[r * 3 + 2] = c2s;
        out_acc[r] = as;
    }
}

extern "C" int hrf_composite_fwd(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                                 const float* background, int64_t num_rays, float step, float* out_color,
                                 float* out_acc, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_start && out_color && out_acc, "NULL argument");
    hipLaunchKernelGGL(k_composite_fwd, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, sigma,
                       (const __half*)rgb, t, ray_start, background, num_rays, step, out_color, out_acc);
    HRF_CHECK_LAUNCH();
    return 0;
}

// composite backward: with g_k = dL/dw_k = sum_ch dC_ch (c_k,ch - bg_ch) + dA,
//   dL/dsigma_i = dt_i ( g_i T_{i+1} - sum_{k>i} g_k w_k ),   dL/dc_i = w_i dC.
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ sigma, const __half* __restrict__ rgb,
                                                       const float* __restrict__ t, const int32_t* __restrict__ ray_start,
                                                       const float* __restrict__ background, const float* __restrict__ d_color,
                                                       const float* __restrict__ d_acc, int64_t num_rays, float step,
                                                       float* __restrict__ d_sigma, float* __restrict__ d_rgb)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    if (b >= e) return;
    const float dc0 = d_color[r * 3 + 0], dc1 = d_color[r * 3 + 1], dc2 = d_color[r * 3 + 2];
    float da = d_acc ? d_acc[r] : 0.0f;
    if (background) da -= dc0 * background[r * 3 + 0] + dc1 * background[r * 3 + 1] + dc2 * background[r * 3 + 2];
    // pass 1: total optical depth of the ray
    float total_sd = 0.0f;
    for (int32_t base = b; base < e; base += 64) {
        const int32_t i = base + lane;
        float sd = 0.0f;
        if (i < e) {
            const float ti = t[i];
            sd = sigma[i] * ((ti + step) - ti);
        }
        total_sd += sd;
    }
    total_sd = wave_sum(total_sd);
    // pass 2, back to front: suffix sums are accumulated directly (a "total - prefix" form cancels badly for the
    // samples near the surface, where the suffix is orders of magnitude smaller than the total).
    float carry_sd = 0.0f, carry_gw = 0.0f;  // sums over all samples behind the current chunk
    const int32_t n_chunks = (e - b + 63) / 64;
    for (int32_t ch = n_chunks - 1; ch >= 0; --ch) {
        const int32_t i = b + ch * 64 + lane;
        float sd = 0.0f, g = 0.0f, dt = 0.0f;
        if (i < e) {
            const float ti = t[i];
            dt = (ti + step) - ti;
            sd = sigma[i] * dt;
            g = dc0 * __half2float(rgb[i * 3 + 0]) + dc1 * __half2float(rgb[i * 3 + 1]) +
                dc2 * __half2float(rgb[i * 3 + 2]) + da;
        }
        const float sd_suf = wave_suffix_scan(sd, lane);              // inclusive: sum_{k >= i in chunk}
        const float T = expf(-fmaxf(total_sd - (carry_sd + sd_suf), 0.0f));  // exclusive prefix of the ray
        const float ex = expf(-sd);
        const float w = (i < e) ? T * (1.0f - ex) : 0.0f;
        const float gw = g * w;
        const float gw_suf = wave_suffix_scan(gw, lane);
        const float suffix = carry_gw + (gw_suf - gw);                 // sum_{k>i} g_k w_k
        if (i < e) {
            d_sigma[i] = dt * (g * (T * ex) - suffix);
            d_rgb[i * 3 + 0] = w * dc0; d_rgb[i * 3 + 1] = w * dc1; d_rgb[i * 3 + 2] = w * dc2;
        }
        carry_sd += __shfl(sd_suf, 0, 64);
        carry_gw += __shfl(gw_suf, 0, 64);
    }
}

extern "C" int hrf_composite_bwd(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                                 const float* background, const float* d_color, const float* d_acc, int64_t num_rays,
                                 float step, float* d_sigma, float* d_rgb, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_start && d_color && d_sigma && d_rgb, "NULL argument");
    hipLaunchKernelGGL(k_composite_bwd, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, sigma,
                       (const __half*)rgb, t, ray_start, background, d_color, d_acc, num_rays, step, d_sigma, d_rgb);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Render + loss + backward of the composite in ONE launch (the fused training step, round 6): per ray -- one wavefront -- the three
// kernels k_composite_fwd, k_loss, k_composite_bwd ran back to back over the same few hundred bytes (volume_rendering.py:123-145,
// trainer.py:205-247 and their autograd), each a launch of its own on the step's critical path with the chip mostly idle (15 + 25 +
// 15 us). Same expressions in the same order as those kernels: colour, opacity, d_sigma, d_rgb are bit-identical
// (tests/test_gpu_parity.py). A wavefront walks several rays (at most 1024 workgroups) and keeps their loss sums in registers; a
// workgroup adds its total to out_sums once -- one add per ray's wavefront would queue 37 000 atomics on three addresses (the first
// version, with a ticket per workgroup on top: 0.31 ms for a kernel of 0.03).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_render_loss(
    const float* __restrict__ sigma, const __half* __restrict__ rgb, const float* __restrict__ t, const int32_t* __restrict__ ray_start,
    const float* __restrict__ background, const float* __restrict__ rgba, int64_t num_rays, int64_t norm_rays, float step, float delta,
    float bce_weight, float grad_scale, const hrf_grad_scaler* __restrict__ scaler, const int32_t* __restrict__ ray_frames,
    const int32_t* __restrict__ f2s, int32_t* __restrict__ group_touched, float* __restrict__ out_color, float* __restrict__ out_acc,
    float* __restrict__ d_sigma, float* __restrict__ d_rgb, float* __restrict__ out_sums)
{
    __shared__ float s_part[4][3];
    if (scaler) grad_scale *= scaler->scale;   // GradScaler.scale(loss), trainer.py:250
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float hub_s = 0.0f, bce_s = 0.0f, se_s = 0.0f;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < num_rays; r += n_waves) {
        float hub = 0.0f, bce = 0.0f, se = 0.0f;
        const int32_t b = ray_start[r], e = ray_start[r + 1];
        // ---- k_composite_fwd
        float carry = 0.0f, c0s = 0.0f, c1s = 0.0f, c2s = 0.0f, as = 0.0f;
        for (int32_t base = b; base < e; base += 64) {
            const int32_t i = base + lane;
            float sd = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
            if (i < e) {
                const float ti = t[i];
                sd = sigma[i] * ((ti + step) - ti);
                cr = __half2float(rgb[i * 3 + 0]); cg = __half2float(rgb[i * 3 + 1]); cb = __half2float(rgb[i * 3 + 2]);
            }
            const float incl = wave_incl_scan(sd, lane);
            const float T = expf(-(carry + (incl - sd)));
            const float w = (i < e) ? T * (1.0f - expf(-sd)) : 0.0f;
            c0s += w * cr; c1s += w * cg; c2s += w * cb; as += w;
            carry += __shfl(incl, 63, 64);
        }
        c0s = wave_sum(c0s); c1s = wave_sum(c1s); c2s = wave_sum(c2s); as = wave_sum(as);
        float bg0 = 1.0f, bg1 = 1.0f, bg2 = 1.0f;
        if (background) {  // volume_rendering.py:144-145
            bg0 = background[r * 3 + 0]; bg1 = background[r * 3 + 1]; bg2 = background[r * 3 + 2];
            const float om = 1.0f - as;
            c0s = c0s + bg0 * om; c1s = c1s + bg1 * om; c2s = c2s + bg2 * om;
        }
        if (lane == 0) {
            if (out_color) { out_color[r * 3 + 0] = c0s; out_color[r * 3 + 1] = c1s; out_color[r * 3 + 2] = c2s; }
            if (out_acc) out_acc[r] = as;
            if (group_touched) {  // plain stores of the same value: no atomics needed
                const int grp = 1 + f2s[ray_frames[r]];
                if (group_touched[grp] == 0) group_touched[grp] = 1;
            }
        }
        // ---- k_loss (every lane computes the ray's scalars)
        const float m = rgba[r * 4 + 3];
        const float inv_n3 = 1.0f / (float)(norm_rays * 3), inv_n = 1.0f / (float)norm_rays;
        const float col[3] = {c0s, c1s, c2s}, bgv[3] = {bg0, bg1, bg2};
        float dc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gt = rgba[r * 4 + k] * m + bgv[k] * (1.0f - m);
            const float ee = col[k] - gt;
            const float ae = fabsf(ee);
            hub += (ae <= delta) ? 0.5f * ee * ee : delta * (ae - 0.5f * delta);
            se += ee * ee;
            const float ge = (ae <= delta) ? ee : (ee > 0.0f ? delta : -delta);
            dc[k] = ge * inv_n3 * grad_scale;
        }
        const float a = as;
        const float pp = fminf(fmaxf(a, 0.0f), 1.0f);
        bce = -(m * logf(pp + 1e-10f) + (1.0f - m) * logf(1.0f - pp + 1e-10f));
        float gp = -(m / (pp + 1e-10f) - (1.0f - m) / (1.0f - pp + 1e-10f));
        if (!(a >= 0.0f && a <= 1.0f)) gp = 0.0f;  // torch.clamp backward mask
        const float dacc = gp * inv_n * bce_weight * grad_scale;
        hub_s += hub; bce_s += bce; se_s += se;
        // ---- k_composite_bwd
        if (b < e) {
            const float dc0 = dc[0], dc1 = dc[1], dc2 = dc[2];
            float da = dacc;
            if (background) da -= dc0 * bg0 + dc1 * bg1 + dc2 * bg2;
            float total_sd = 0.0f;
            for (int32_t base = b; base < e; base += 64) {
                const int32_t i = base + lane;
                float sd = 0.0f;
                if (i < e) {
                    const float ti = t[i];
                    sd = sigma[i] * ((ti + step) - ti);
                }
                total_sd += sd;
            }
            total_sd = wave_sum(total_sd);
            float carry_sd = 0.0f, carry_gw = 0.0f;
            const int32_t n_chunks = (e - b + 63) / 64;
            for (int32_t ch = n_chunks - 1; ch >= 0; --ch) {
                const int32_t i = b + ch * 64 + lane;
                float sd = 0.0f, g = 0.0f, dt = 0.0f;
                if (i < e) {
                    const float ti = t[i];
                    dt = (ti + step) - ti;
                    sd = sigma[i] * dt;
                    g = dc0 * __half2float(rgb[i * 3 + 0]) + dc1 * __half2float(rgb[i * 3 + 1]) +
                        dc2 * __half2float(rgb[i * 3 + 2]) + da;
                }
                const float sd_suf = wave_suffix_scan(sd, lane);
                const float T = expf(-fmaxf(total_sd - (carry_sd + sd_suf), 0.0f));
                const float ex = expf(-sd);
                const float w = (i < e) ? T * (1.0f - ex) : 0.0f;
                const float gw = g * w;
                const float gw_suf = wave_suffix_scan(gw, lane);
                const float suffix = carry_gw + (gw_suf - gw);
                if (i < e) {
                    d_sigma[i] = dt * (g * (T * ex) - suffix);
                    d_rgb[i * 3 + 0] = w * dc0; d_rgb[i * 3 + 1] = w * dc1; d_rgb[i * 3 + 2] = w * dc2;
                }
                carry_sd += __shfl(sd_suf, 0, 64);
                carry_gw += __shfl(gw_suf, 0, 64);
            }
        }
    }
    if (out_sums) {   // the workgroup's rays -> three adds
        if (lane == 0) { s_part[wave][0] = hub_s; s_part[wave][1] = bce_s; s_part[wave][2] = se_s; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const float v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
            if (v != 0.0f) unsafeAtomicAdd(out_sums + threadIdx.x, v);
        }
    }
}

extern "C" int hrf_render_loss_fused(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                                     const float* background, const float* rgba, int64_t num_rays, int64_t norm_rays, float step,
                                     float huber_delta, float bce_weight, float grad_scale, const hrf_grad_scaler* scaler,
                                     const int32_t* ray_frames, const int32_t* frame_to_segment, int32_t* group_touched,
                                     float* out_color, float* out_acc, float* d_sigma, float* d_rgb, float* out_sums,
                                     hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_start && rgba && d_sigma && d_rgb, "NULL argument");
    HRF_CHECK_ARG(norm_rays == 0 || norm_rays >= num_rays, "norm_rays smaller than the rays of this call");
    HRF_CHECK_ARG(!group_touched || (ray_frames && frame_to_segment), "group flags requested without frames");
    if (norm_rays == 0) norm_rays = num_rays;
    unsigned blocks = hrf_blocks(num_rays * 64, 256);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(k_render_loss, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sigma, (const __half*)rgb, t, ray_start,
                       background, rgba, num_rays, norm_rays, step, huber_delta, bce_weight, grad_scale, scaler, ray_frames,
                       frame_to_segment, group_touched, out_color, out_acc, d_sigma, d_rgb, out_sums);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone forms of the three nerfacc 0.3.1 calls of volume_rendering.py:75-81,123-141 (the training step uses the
// fused composite above; these exist so that code written against nerfacc's functions keeps working).
// render_weight_from_density: w_i = T_i (1 - exp(-sigma_i dt_i)), T_i = exp(-sum_{j<i} sigma_j dt_j), dt = t_end - t_start.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weights_fwd(const float* __restrict__ sigma, const float* __restrict__ t0,
                                                     const float* __restrict__ t1, const int32_t* __restrict__ ray_start,
                                                     int64_t num_rays, float* __restrict__ w_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    float carry = 0.0f;
    for (int32_t base = b; base < e; base += 64) {
        const int32_t i = base + lane;
        const float sd = (i < e) ? sigma[i] * (t1[i] - t0[i]) : 0.0f;
        const float incl = wave_incl_scan(sd, lane);
        const float T = expf(-(carry + (incl - sd)));
        if (i < e) w_out[i] = T * (1.0f - expf(-sd));
        carry += __shfl(incl, 63, 64);
    }
}

// dL/dsigma_i = dt_i ( g_i T_{i+1} - sum_{k>i} g_k w_k ),  g = dL/dw  (back to front, as in k_composite_bwd)
__global__ __launch_bounds__(256) void k_weights_bwd(const float* __restrict__ sigma, const float* __restrict__ t0,
                                                     const float* __restrict__ t1, const int32_t* __restrict__ ray_start,
                                                     const float* __restrict__ d_w, int64_t num_rays,
                                                     float* __restrict__ d_sigma)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    if (b >= e) return;
    float total_sd = 0.0f;
    for (int32_t base = b; base < e; base += 64) {
        const int32_t i = base + lane;
        total_sd += (i < e) ? sigma[i] * (t1[i] - t0[i]) : 0.0f;
    }
    total_sd = wave_sum(total_sd);
    float carry_sd = 0.0f, carry_gw = 0.0f;
    const int32_t n_chunks = (e - b + 63) / 64;
    for (int32_t ch = n_chunks - 1; ch >= 0; --ch) {
        const int32_t i = b + ch * 64 + lane;
        float sd = 0.0f, g = 0.0f, dt = 0.0f;
        if (i < e) { dt = t1[i] - t0[i]; sd = sigma[i] * dt; g = d_w[i]; }
        const float sd_suf = wave_suffix_scan(sd, lane);
        const float T = expf(-fmaxf(total_sd - (carry_sd + sd_suf), 0.0f));
        const float ex = expf(-sd);
        const float w = (i < e) ? T * (1.0f - ex) : 0.0f;
        const float gw = g * w;
        const float gw_suf = wave_suffix_scan(gw, lane);
        if (i < e) d_sigma[i] = dt * (g * (T * ex) - (carry_gw + (gw_suf - gw)));
        carry_sd += __shfl(sd_suf, 0, 64);
        carry_gw += __shfl(gw_suf, 0, 64);
    }
}

extern "C" int hrf_weights_fwd(const float* sigma, const float* t_starts, const float* t_ends, const int32_t* ray_start,
                               int64_t num_rays, float* out_weights, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(sigma && t_starts && t_ends && ray_start && out_weights, "NULL argument");
    hipLaunchKernelGGL(k_weights_fwd, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, sigma, t_starts,
                       t_ends, ray_start, num_rays, out_weights);
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_weights_bwd(const float* sigma, const float* t_starts, const float* t_ends, const int32_t* ray_start,
                               const float* d_weights, int64_t num_rays, float* d_sigma, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(sigma && t_starts && t_ends && ray_start && d_weights && d_sigma, "NULL argument");
    hipLaunchKernelGGL(k_weights_bwd, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, sigma, t_starts,
                       t_ends, ray_start, d_weights, num_rays, d_sigma);
    HRF_CHECK_LAUNCH();
    return 0;
}

// accumulate_along_rays: out[r][d] = sum_{i in ray r} w_i * v_i[d]  (values == NULL: v = 1, D = 1). One wavefront per ray,
// no atomics (samples are sorted by ray).
__global__ __launch_bounds__(256) void k_accumulate_fwd(const float* __restrict__ w, const float* __restrict__ values, int D,
                                                        const int32_t* __restrict__ ray_start, int64_t num_rays,
                                                        float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    const int32_t b = ray_start[r], e = ray_start[r + 1];
    for (int d = 0; d < D; ++d) {
        float acc = 0.0f;
        for (int32_t i = b + lane; i < e; i += 64) acc += w[i] * (values ? values[(size_t)i * D + d] : 1.0f);
        acc = wave_sum(acc);
        if (lane == 0) out[r * D + d] = acc;
    }
}

__global__ __launch_bounds__(256) void k_accumulate_bwd(const float* __restrict__ w, const float* __restrict__ values, int D,
                                                        const int64_t* __restrict__ sample_ray, const float* __restrict__ d_out,
                                                        int64_t n, float* __restrict__ d_w, float* __restrict__ d_values)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = sample_ray[i];
    const float wi = w[i];
    float g = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float go = d_out[r * D + d];
        g += go * (values ? values[(size_t)i * D + d] : 1.0f);
        if (d_values) d_values[(size_t)i * D + d] = wi * go;
    }
    if (d_w) d_w[i] = g;
}

extern "C" int hrf_accumulate_fwd(const float* weights, const float* values, int value_dim, const int32_t* ray_start,
                                  int64_t num_rays, float* out, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(weights && ray_start && out && value_dim >= 1 && value_dim <= 64, "bad argument");
    hipLaunchKernelGGL(k_accumulate_fwd, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, weights,
                       values, value_dim, ray_start, num_rays, out);
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_accumulate_bwd(const float* weights, const float* values, int value_dim, const int64_t* sample_ray,
                                  const float* d_out, int64_t n, float* d_weights, float* d_values, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(weights && sample_ray && d_out && value_dim >= 1 && value_dim <= 64, "bad argument");
    hipLaunchKernelGGL(k_accumulate_bwd, dim3(hrf_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, weights, values,
                       value_dim, sample_ray, d_out, n, d_weights, d_values);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// loss (trainer.py:205-247): gt = rgb*mask + bg*(1-mask); Huber(delta, mean over R*3) +
// bce_weight * mean BCE(clamp(acc,0,1), mask) (utils/loss.py:4-10). Gradients are multiplied by grad_scale (x the device-side GradScaler's scale).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss(const float* __restrict__ color, const float* __restrict__ acc,
                                              const float* __restrict__ rgba, const float* __restrict__ background,
                                              int64_t num_rays, int64_t norm_rays, float delta, float bce_weight,
                                              float grad_scale, float* __restrict__ d_color, float* __restrict__ d_acc,
                                              float* __restrict__ out_sums, const int32_t* __restrict__ ray_frames,
                                              const int32_t* __restrict__ f2s, int32_t* __restrict__ group_touched,
                                              const hrf_grad_scaler* __restrict__ scaler)
{
    if (scaler) grad_scale *= scaler->scale;   // GradScaler.scale(loss), trainer.py:250
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float hub = 0.0f, bce = 0.0f, se = 0.0f;
    if (r < num_rays) {
        if (group_touched) {  // plain stores of the same value: no atomics needed
            const int grp = 1 + f2s[ray_frames[r]];
            if (group_touched[grp] == 0) group_touched[grp] = 1;
        }
        const float m = rgba[r * 4 + 3];
        const float inv_n3 = 1.0f / (float)(norm_rays * 3), inv_n = 1.0f / (float)norm_rays;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float bg = background ? background[r * 3 + k] : 1.0f;
            const float gt = rgba[r * 4 + k] * m + bg * (1.0f - m);
            const float e = color[r * 3 + k] - gt;
            const float ae = fabsf(e);
            hub += (ae <= delta) ? 0.5f * e * e : delta * (ae - 0.5f * delta);
            se += e * e;
            const float ge = (ae <= delta) ? e : (e > 0.0f ? delta : -delta);
            d_color[r * 3 + k] = ge * inv_n3 * grad_scale;
        }
        const float a = acc[r];
        const float p = fminf(fmaxf(a, 0.0f), 1.0f);
        bce = -(m * logf(p + 1e-10f) + (1.0f - m) * logf(1.0f - p + 1e-10f));
        float gp = -(m / (p + 1e-10f) - (1.0f - m) / (1.0f - p + 1e-10f));
        if (!(a >= 0.0f && a <= 1.0f)) gp = 0.0f;  // torch.clamp backward mask
        d_acc[r] = gp * inv_n * bce_weight * grad_scale;
    }
    hub = wave_sum(hub); bce = wave_sum(bce); se = wave_sum(se);
    if ((threadIdx.x & 63) == 0 && out_sums) {
        unsafeAtomicAdd(out_sums + 0, hub);
        unsafeAtomicAdd(out_sums + 1, bce);
        unsafeAtomicAdd(out_sums + 2, se);
    }
}

extern "C" int hrf_loss_fwd_bwd(const float* color, const float* acc, const float* rgba, const float* background,
                                int64_t num_rays, int64_t norm_rays, float huber_delta, float bce_weight, float grad_scale,
                                float* d_color, float* d_acc, float* out_sums, const int32_t* ray_frames,
                                const int32_t* frame_to_segment, int32_t* group_touched, const hrf_grad_scaler* scaler,
                                hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(color && acc && rgba && d_color && d_acc, "NULL argument");
    HRF_CHECK_ARG(norm_rays == 0 || norm_rays >= num_rays, "norm_rays smaller than the rays of this call");
    if (norm_rays == 0) norm_rays = num_rays;
    HRF_CHECK_ARG(!group_touched || (ray_frames && frame_to_segment), "group flags requested without frames");
    hipLaunchKernelGGL(k_loss, dim3(hrf_blocks(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, color, acc, rgba,
                       background, num_rays, norm_rays, huber_delta, bce_weight, grad_scale, d_color, d_acc, out_sums, ray_frames,
                       frame_to_segment, group_touched, scaler);
    HRF_CHECK_LAUNCH();
    return 0;
}
