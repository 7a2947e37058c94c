// Fused pruning pass for gfx950: per-ray march with early termination.
//
// Replaces the body of prune_samples (humanrf/volume_rendering.py:63-84) -- jitter, positions, HumanRF.density
// (4 hash grids + compose + sigma_net + truncated_exp, humanrf.py:158-186), alpha = 1 - exp(-sigma*step) and
// nerfacc.render_visibility -- with ONE kernel that never evaluates a sample the reference would discard for
// transmittance: visibility is (T_i >= eps) && (alpha_i >= thre) with T non-increasing along the ray, so once
// T drops below eps every later sample is invisible whatever its density. The reference
// encodes all N0 occupancy-surviving samples (~250 per ray) to keep ~40; marching stops after the chunk in
// which the ray saturates. Results are identical to the unfused path (same arithmetic, same sequential fp32
// transmittance product), only the work differs.
//
// One wavefront per ray; per iteration the 64 lanes take 64 consecutive samples of the ray's run:
//   encode (16 levels x 4 encodings x 8 corners per lane, segment metadata is wavefront-uniform)
//   -> features through a wavefront-private LDS tile into MFMA B fragments -> sigma_net (4 tiles of 16 samples)
//   -> alpha, sequential transmittance via v_readlane, __ballot + prefix popcount compaction of the survivors
//   into the ray's own slot range. A later pass packs the ranges (hrf_pack_runs).
#include "encode_common.h"
#include "mlp_common.h"
#include <algorithm>
#include <cstdlib>

#ifndef MARCH_ROW
#define MARCH_ROW 40  // halves per LDS feature row (32 + 8 pad: 80-byte rows, 8-byte aligned fragments)
#endif

// The 64 lanes of the wavefront always work on 64 consecutive samples of ONE ray (shorter steps of 16 / 32 samples
// were measured: they save < 15 % of the encoded samples and cost more per step). Per-ray constants are hoisted:
// the time coordinate is the same for every sample of a ray, so the interpolated time vector is computed once per ray
// (lane f holds feature f, v_readlane hands it out), and the three spatial vectors are fetched for two levels at a
// time (one 16-byte load per tap instead of two 8-byte loads: the taps of a step touch ~50 distinct rows, so their
// cost is the number of load instructions, not the bytes).
template <class P>
#ifndef MARCH_WAVES
#define MARCH_WAVES 4   // wavefronts per SIMD the register allocation aims at (128 VGPRs)
#endif
__global__ __launch_bounds__(128, MARCH_WAVES) void k_prune_march(
    const float* __restrict__ ray_o, const float* __restrict__ ray_d, const int32_t* __restrict__ ray_frames,
    const int32_t* __restrict__ ray_start, const float* __restrict__ t0, const float* __restrict__ jitter, float step,
    float eps, float thre, const int32_t* __restrict__ f2s, const float* __restrict__ f2l,
    const __half2* __restrict__ tables, const float* __restrict__ vectors, const hrf_segment_meta* __restrict__ segs,
    int vec_res, const typename P::E* __restrict__ w1, const typename P::E* __restrict__ w2, float density_scale, int64_t num_rays,
    const int32_t* __restrict__ num_rays_dev, int64_t capacity, float* __restrict__ t_stage,
    float* __restrict__ sigma_stage, int32_t* __restrict__ ray_cnt, int32_t* __restrict__ ray_evaluated,
    const int32_t* __restrict__ ray_order, const int32_t* __restrict__ ray_len, uint32_t jitter_seed,
    unsigned long long* __restrict__ totals, int phase_shift)
{
    constexpr int CH = 64;
    typedef typename P::V V;
    __shared__ __attribute__((aligned(16))) typename P::E s_w1[64 * (32 + WPAD)];
    __shared__ __attribute__((aligned(16))) typename P::E s_w2[16 * (64 + WPAD)];
    __shared__ __attribute__((aligned(16))) _Float16 s_feat[2][CH * MARCH_ROW];
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const unsigned long long le_mask = (2ull << lane) - 1ull;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wavefront-uniform: per-ray state in SGPRs
    _Float16* feat = s_feat[wv];
    const int n_rays = (int)num_rays;
    const int live_rays = num_rays_dev ? min(*num_rays_dev, n_rays) : n_rays;
    // host passed an upper bound: slots beyond the device-side count hold nothing
    for (int r = live_rays + (int)blockIdx.x * 2 + wv; r < n_rays; r += (int)gridDim.x * 2)
        if (lane == 0) { ray_cnt[r] = 0; if (ray_evaluated) ray_evaluated[r] = 0; }
    // XCD-aware ray assignment: workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8),
    // each XCD has its own 4 MB L2, and a ray only reads the tables of its own temporal segment. XCD x therefore takes
    // the x-th eighth of the rays IN SEGMENT ORDER (ray_order = ray ids sorted by segment, hrf_ray_segment_order), so
    // one L2 sees one or two segments (8-16 MB of tables) instead of all of them (74 MB at 50 frames). The launcher
    // rounds the grid up to a multiple of 8.
    const int xcd = (int)(blockIdx.x & 7u);
    const int chunk = (live_rays + 7) >> 3;
    const int waves_per_xcd = (int)(gridDim.x >> 3) * 2;
    const int p_end = min(live_rays, (xcd + 1) * chunk);
    // The grid is sized for the host's upper bound (the DRAWN rays); ~10 % of them survive the occupancy mask, so most
    // workgroups own no ray: they leave before staging the weights (workgroup-uniform test, the barrier below is safe).
    if (xcd * chunk + (int)(blockIdx.x >> 3) * 2 >= p_end) return;
    // Two wavefronts per workgroup: rays differ a lot in length, so small workgroups let the dispatcher balance;
    // two (not one) keeps the LDS footprint per wavefront low enough for 4 wavefronts per SIMD.
    stage_rm(s_w1, w1, 64, 32);
    stage_rm(s_w2, w2, 16, 64);
    __syncthreads();
    for (int p = xcd * chunk + (int)(blockIdx.x >> 3) * 2 + wv; p < p_end; p += waves_per_xcd) {
        const int r = ray_order ? __builtin_amdgcn_readfirstlane(ray_order[p]) : p;
        // the ray's staged samples: [ray_start[r], ray_start[r+1]), or a prefix of that range when the sampler staged by
        // candidate count in one pass (ray_len = samples that passed the occupancy predicate)
        const int32_t rb = ray_start[r], re = ray_len ? rb + ray_len[r] : ray_start[r + 1];
        const float ox = ray_o[(size_t)r * 3 + 0], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
        const float dx = ray_d[r * 3 + 0], dy = ray_d[r * 3 + 1], dz = ray_d[r * 3 + 2];
        const int32_t frame = ray_frames[r];
        const int seg = __builtin_amdgcn_readfirstlane(f2s[frame]);
        const float tl = f2l[frame];
        const hrf_segment_meta* sm = segs + seg;
        const __half2* tbase = tables + sm->table_offset;
        const uint32_t entries = sm->entries;
        const float* vbase = vectors + (size_t)seg * 4 * vec_res * ENC_F;
        // time vector (index 3), interpolated once per ray: lane f (and f + 32) holds feature f
        int vt_lane;
        {
            int c0, c1;
            float fr;
            hrf_vec_tap(tl, vec_res, c0, c1, fr);
            const float v0 = vbase[((size_t)3 * vec_res + c0) * ENC_F + (lane & 31)];
            const float v1 = vbase[((size_t)3 * vec_res + c1) * ENC_F + (lane & 31)];
            vt_lane = __builtin_bit_cast(int, v0 + fr * (v1 - v0));
        }
        float T = 1.0f;
        int32_t kept = 0, evaluated = 0;
        for (int32_t cb = rb; cb < re; cb += CH) {
            const int32_t i = cb + lane;
            const bool valid = i < re && i < capacity;  // capacity: size of t0 / jitter / t_stage
            float t = 0.0f;
            if (valid) {
                t = t0[i];
                if (jitter) t = t + jitter[i] * step;  // volume_rendering.py:63-64
                else if (jitter_seed) t = t + hrf_uniform01(jitter_seed, (uint32_t)i) * step;
            }
            EncCoords q;
            q.c[0] = (ox + t * dx) + 0.5f;  // volume_rendering.py:68-69, humanrf.py:175
            q.c[1] = (oy + t * dy) + 0.5f;
            q.c[2] = (oz + t * dz) + 0.5f;
            q.c[3] = tl;
            int vo0[3], vo1[3];  // float offsets of the two tap rows of the x / y / z vectors
            float vfr[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int c0, c1;
                hrf_vec_tap(q.c[k], vec_res, c0, c1, vfr[k]);
                vo0[k] = (k * vec_res + c0) * ENC_F;
                vo1[k] = (k * vec_res + c1) * ENC_F;
            }
            // Models with fewer than 16 levels (n_levels < 16, model_args.py:16): sigma_net's input is the 2 n_levels features,
            // padded by tcnn to a multiple of 16 with ONES (the Identity encoding's padding, [UPSTREAM-KNOWLEDGE]); the kernels'
            // rows stay 32 wide, the columns beyond the padding hold zeros (their weights never matter).
            const int n_lv = (int)sm->n_levels, ones_end = (2 * n_lv + 15) & ~15;
            uint32_t lp_done = 0;
#pragma unroll 1
            for (int lpk = 0; lpk < 8; ++lpk) {  // two levels (four features) per iteration
                const int lp = enc_phase_next<8>(lpk, phase_shift, lp_done);   // (order: see enc_phase_next)
                if (2 * lp >= n_lv) {          // (wave-uniform) no level left in this pair
                    const float c0 = 4 * lp < ones_end ? 1.0f : 0.0f, c1 = 4 * lp + 2 < ones_end ? 1.0f : 0.0f;
                    *(__half2*)(feat + lane * MARCH_ROW + 4 * lp) = __floats2half2_rn(c0, c0);
                    *(__half2*)(feat + lane * MARCH_ROW + 4 * lp + 2) = __floats2half2_rn(c1, c1);
                    continue;
                }
                float sv[3][4];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 v0 = *(const float4*)(vbase + vo0[k] + 4 * lp);
                    const float4 v1 = *(const float4*)(vbase + vo1[k] + 4 * lp);
                    sv[k][0] = v0.x + vfr[k] * (v1.x - v0.x);
                    sv[k][1] = v0.y + vfr[k] * (v1.y - v0.y);
                    sv[k][2] = v0.z + vfr[k] * (v1.z - v0.z);
                    sv[k][3] = v0.w + vfr[k] * (v1.w - v0.w);
                }
#pragma unroll 1
                for (int j = 0; j < 2; ++j) {  // rolled: one level's 32 gathers in flight at a time (register budget)
                    const int l = 2 * lp + j;
                    if (l >= n_lv) {   // (wave-uniform) odd n_levels: the second level of the last pair
                        const float cc = 2 * l < ones_end ? 1.0f : 0.0f;
                        *(__half2*)(feat + lane * MARCH_ROW + 2 * l) = __floats2half2_rn(cc, cc);
                        continue;
                    }
                    const hrf_level_meta lv = sm->levels[l];
                    float fe[4][2];
#ifdef MARCH_PLAIN_FROM_LEVEL
                    if (l >= MARCH_PLAIN_FROM_LEVEL) enc_level_plain(q, tbase, entries, lv, fe);   // (wave-uniform)
                    else
#endif
                    ENC_LEVEL_SHARED(q, tbase, entries, lv, le_mask, fe);
                    const float st0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vt_lane, 2 * l));
                    const float st1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vt_lane, 2 * l + 1));
                    const float sx0 = j ? sv[0][2] : sv[0][0], sx1 = j ? sv[0][3] : sv[0][1];
                    const float sy0 = j ? sv[1][2] : sv[1][0], sy1 = j ? sv[1][3] : sv[1][1];
                    const float sz0 = j ? sv[2][2] : sv[2][0], sz1 = j ? sv[2][3] : sv[2][1];
                    // tensor_composition.cu:47-54: xyz*v_t + xyt*v_z + yzt*v_x + xzt*v_y
                    const float r0 = ((fe[0][0] * st0 + fe[1][0] * sz0) + fe[2][0] * sx0) + fe[3][0] * sy0;
                    const float r1 = ((fe[0][1] * st1 + fe[1][1] * sz1) + fe[2][1] * sx1) + fe[3][1] * sy1;
                    *(__half2*)(feat + lane * MARCH_ROW + 2 * l) = __floats2half2_rn(r0, r1);
                }
            }
            // wavefront-private LDS: no barrier, only ordering of this wavefront's own DS operations
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float sigma = 0.0f;
            // weight fragments are re-read from LDS per step (8-byte reads) instead of living in 24 VGPRs across
            // the gather phase: keeps the kernel at 4 wavefronts per SIMD
            V a1[4][2], a2[4];
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                a1[ht][0] = afrag(s_w1, 32, ht, 0, lane);
                a1[ht][1] = afrag(s_w1, 32, ht, 1, lane);
                a2[ht] = afrag(s_w2, 64, 0, ht, lane);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const V x0 = pv_from_h4<P>(*(const h4*)(feat + (16 * tt + c) * MARCH_ROW + 4 * g));
                const V x1 = pv_from_h4<P>(*(const h4*)(feat + (16 * tt + c) * MARCH_ROW + 16 + 4 * g));
                V hid[4];
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) hid[ht] = pv_relu<P>(P::mfma2(a1[ht][0], a1[ht][1], x0, x1, f4zero()));
                f4 o = P::mfma2(a2[0], a2[1], hid[0], hid[1], f4zero());   // v_mfma_f32_16x16x32: 6 instead of 12 per 16 samples
                o = P::mfma2(a2[2], a2[3], hid[2], hid[3], o);
                // h0 of sample 16*tt + c sits in lane c (g == 0): hand it to every lane whose sample that is
                // (rounded to the network's type, then to the fp16 container k_density_fwd stores it in)
                const float h0 = hround(p_round<P>(o[0]));
                const float mine = __shfl(h0, lane & 15, 64);
                if ((lane >> 4) == tt) sigma = expf(mine) * density_scale;  // truncated_exp forward, humanrf.py:184
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // visibility (same sequential product as k_visibility / oracle orc_visibility)
            const float a = valid ? (1.0f - expf(-sigma * step)) : 0.0f;  // volume_rendering.py:76
            const float om = 1.0f - a;
            float myT = 0.0f;
            const int cnt = min(CH, re - cb);
            for (int k = 0; k < cnt; ++k) {
                if (lane == k) myT = T;
                T = T * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, om), k));
            }
            const bool vis = valid && (myT >= eps) && (a >= thre);
            const unsigned long long bal = __ballot(vis);
            if (vis) {
                const int pre = __popcll(bal & ((1ull << lane) - 1ull));
                t_stage[rb + kept + pre] = t;
                if (sigma_stage) sigma_stage[rb + kept + pre] = sigma;
            }
            kept += __popcll(bal);
            evaluated += cnt;
            if (T < eps) break;
        }
        if (lane == 0) {
            ray_cnt[r] = kept;
            if (ray_evaluated) ray_evaluated[r] = evaluated;
            if (totals) {  // statistics: samples handed to the pruning pass / samples it encoded (non-returning atomics)
                atomicAdd(&totals[0], (unsigned long long)(re - rb));
                atomicAdd(&totals[1], (unsigned long long)evaluated);
            }
        }
    }
}

extern "C" int hrf_prune_march(const float* ray_origins, const float* ray_dirs, const int32_t* ray_frames,
                               const int32_t* ray_start, const float* t0, const float* jitter, float step,
                               float early_stop_eps, float alpha_thre, const int32_t* frame_to_segment,
                               const float* frame_to_local, const void* tables, const float* vectors,
                               const hrf_segment_meta* segments, int num_segments, int vec_res, const void* w1,
                               const void* w2, float density_scale, int64_t num_rays, const int32_t* num_rays_dev,
                               int64_t capacity, float* t_stage, float* sigma_stage, int32_t* ray_cnt, int32_t* ray_evaluated,
                               const int32_t* ray_order, const int32_t* ray_len, uint32_t jitter_seed,
                               uint64_t* totals, int mlp_bf16, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_origins && ray_dirs && ray_frames && ray_start && t0, "NULL ray / sample input");
    HRF_CHECK_ARG(frame_to_segment && frame_to_local && tables && vectors && segments && w1 && w2, "NULL model input");
    HRF_CHECK_ARG(t_stage && ray_cnt, "NULL output");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1, "bad segment count / vector resolution");
    HRF_CHECK_ARG(num_rays < (int64_t)1 << 29, "too many rays for one launch");
    HRF_CHECK_ARG(!(jitter && jitter_seed), "pass either a jitter array or a jitter seed, not both");
    unsigned blocks = (unsigned)std::min<int64_t>((num_rays + 1) / 2, 1 << 20);
    blocks = (blocks + 7u) & ~7u;  // whole rounds over the 8 XCDs
    int phase_shift = ENC_PHASE_SHIFT;   // (see enc_phase_next, encode_common.h)
#ifdef ENC_PHASE_TUNE
    if (const char* e = getenv("HRF_PHASE_SHIFT")) phase_shift = atoi(e);
#endif
#define HRF_LAUNCH_PM(PP, ET)                                                                                          \
    hipLaunchKernelGGL(k_prune_march<PP>, dim3(blocks), dim3(128), 0, (hipStream_t)stream, ray_origins, ray_dirs,      \
                       ray_frames, ray_start, t0, jitter, step, early_stop_eps, alpha_thre, frame_to_segment,          \
                       frame_to_local, (const __half2*)tables, vectors, segments, vec_res, (const ET*)w1, (const ET*)w2, \
                       density_scale, num_rays, num_rays_dev, capacity, t_stage, sigma_stage, ray_cnt, ray_evaluated,  \
                       ray_order, ray_len, jitter_seed, (unsigned long long*)totals, phase_shift)
    if (mlp_bf16) HRF_LAUNCH_PM(Prec<true>, short); else HRF_LAUNCH_PM(Prec<false>, _Float16);
#undef HRF_LAUNCH_PM
    HRF_CHECK_LAUNCH();
    return 0;
}

// Ray ids sorted by a per-frame key -- the temporal segment, or the frame itself (rays of one frame also share the
// time slice of the three space-time tables) -- by counting sort, order inside a key arbitrary: the schedule of the
// march. workspace: 2 * num_keys int32 (histogram, cursors), zeroed by the launcher.
__global__ __launch_bounds__(256) void k_segment_hist(const int32_t* __restrict__ ray_frames, const int32_t* __restrict__ f2s,
                                                      int num_rays, const int32_t* __restrict__ num_rays_dev,
                                                      int num_keys, int32_t* __restrict__ hist)
{
    __shared__ int32_t s_cnt[1024];
    const int live = num_rays_dev ? min(*num_rays_dev, num_rays) : num_rays;
    if ((int)(blockIdx.x * blockDim.x) >= live) return;  // whole workgroup beyond the device-side count
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < live) atomicAdd(&s_cnt[f2s[ray_frames[r]]], 1);  // LDS: global atomics on a handful of hot counters serialise
    __syncthreads();
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&hist[k], s_cnt[k]);
}

__global__ __launch_bounds__(256) void k_segment_scatter(const int32_t* __restrict__ ray_frames, const int32_t* __restrict__ f2s,
                                                         int num_rays, const int32_t* __restrict__ num_rays_dev,
                                                         int num_keys, const int32_t* __restrict__ hist,
                                                         int32_t* __restrict__ cursor, int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ values, int32_t* __restrict__ out_values)
{
    __shared__ int32_t s_cnt[1024];   // rays of this workgroup per key, then the workgroup's base inside the key's range
    const int live = num_rays_dev ? min(*num_rays_dev, num_rays) : num_rays;
    if ((int)(blockIdx.x * blockDim.x) >= live) return;
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int key = -1, rank = 0;
    if (r < live) {
        key = f2s[ray_frames[r]];
        rank = atomicAdd(&s_cnt[key], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) {
        const int32_t c = s_cnt[k];
        if (c) {
            int32_t off = 0;  // start of key k in the sorted order: exclusive prefix of the histogram
            for (int q = 0; q < k; ++q) off += hist[q];
            s_cnt[k] = off + atomicAdd(&cursor[k], c);
        }
    }
    __syncthreads();
    if (key >= 0) {
        const int pos = s_cnt[key] + rank;
        order[pos] = r;
        if (values) out_values[pos] = values[r];   // a per-ray value carried into the sorted order (visible-sample counts)
    }
}

extern "C" int hrf_ray_segment_order_values(const int32_t* ray_frames, const int32_t* frame_to_segment, int64_t num_rays,
                                            const int32_t* num_rays_dev, int num_segments, int32_t* workspace,
                                            int32_t* out_order, const int32_t* values, int32_t* out_values,
                                            hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_frames && frame_to_segment && workspace && out_order, "NULL argument");
    HRF_CHECK_ARG(!values == !out_values, "values and out_values go together");
    HRF_CHECK_ARG(num_segments > 0 && num_segments <= 1024, "key count must be in [1,1024]");
    HRF_CHECK_ARG(num_rays < (int64_t)1 << 29, "too many rays for one launch");
    if (hipMemsetAsync(workspace, 0, sizeof(int32_t) * 2 * (size_t)num_segments, (hipStream_t)stream) != hipSuccess) {
        hrf_set_error("%s: hipMemsetAsync failed", __func__);
        return 2;
    }
    const unsigned blocks = (unsigned)((num_rays + 255) / 256);
    hipLaunchKernelGGL(k_segment_hist, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ray_frames, frame_to_segment,
                       (int)num_rays, num_rays_dev, num_segments, workspace);
    hipLaunchKernelGGL(k_segment_scatter, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ray_frames, frame_to_segment,
                       (int)num_rays, num_rays_dev, num_segments, workspace, workspace + num_segments, out_order, values,
                       out_values);
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_ray_segment_order(const int32_t* ray_frames, const int32_t* frame_to_segment, int64_t num_rays,
                                     const int32_t* num_rays_dev, int num_segments, int32_t* workspace,
                                     int32_t* out_order, hrf_stream_t stream)
{
    return hrf_ray_segment_order_values(ray_frames, frame_to_segment, num_rays, num_rays_dev, num_segments, workspace,
                                        out_order, nullptr, nullptr, stream);
}

// Pack the per-ray survivor ranges [ray_start[r], ray_start[r] + ray_cnt[r]) of a staged array into the dense,
// ray-sorted output (volume_rendering.py:83-84): out_offset = exclusive scan of ray_cnt. One wavefront per ray.
__global__ __launch_bounds__(256) void k_pack_runs(const int32_t* __restrict__ ray_start, const int32_t* __restrict__ ray_cnt,
                                                   const int32_t* __restrict__ out_offset, const float* __restrict__ t_stage,
                                                   int64_t num_rays, const int32_t* __restrict__ num_rays_dev,
                                                   int64_t ray_base, float* __restrict__ out_t, int64_t* __restrict__ out_ray)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays || (num_rays_dev && r >= *num_rays_dev)) return;
    const int32_t src = ray_start[r], n = ray_cnt[r], dst = out_offset[r];
    for (int32_t j = lane; j < n; j += 64) {
        out_t[dst + j] = t_stage[src + j];
        out_ray[dst + j] = r + ray_base;  // ray_base: re-basing of humanrf/input.py:24-31 when batches are merged
    }
}

extern "C" int hrf_pack_runs(const int32_t* ray_start, const int32_t* ray_cnt, const int32_t* out_offset,
                             const float* t_stage, int64_t num_rays, const int32_t* num_rays_dev, int64_t ray_base,
                             float* out_t, int64_t* out_ray, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_start && ray_cnt && out_offset && t_stage && out_t && out_ray, "NULL argument");
    hipLaunchKernelGGL(k_pack_runs, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, ray_start,
                       ray_cnt, out_offset, t_stage, num_rays, num_rays_dev, ray_base, out_t, out_ray);
    HRF_CHECK_LAUNCH();
    return 0;
}

// The same packing in a caller-given ray ORDER (the rays of a training batch sorted by frame): sorted ray i is source ray
// order[i]; its visible samples go to out_offset_sorted[i] (exclusive scan of the counts in sorted order) with ray id i,
// and lane 0 of the wavefront moves the ray's own record. A training batch is a set of i.i.d. rays (data_loader.py:540-546)
// and every consumer (render, the loss means, the gradient sums) is invariant under a permutation of its rays; in frame
// order the 64 samples of an encode workgroup and the 1024 of a scatter tile read ONE temporal segment's tables, and an XCD
// that takes a contiguous eighth of the batch keeps one or two frames' tables in its L2 (as the prune march's schedule does).
__global__ __launch_bounds__(256) void k_pack_runs_sorted(
    const int32_t* __restrict__ order, const int32_t* __restrict__ ray_start, const int32_t* __restrict__ ray_cnt,
    const int32_t* __restrict__ out_offset, const float* __restrict__ t_stage, int64_t num_rays,
    const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ rgba,
    const int32_t* __restrict__ frames, const int32_t* __restrict__ cams, const float* __restrict__ minmax,
    const int64_t* __restrict__ pixel, float* __restrict__ o_origins, float* __restrict__ o_dirs, float* __restrict__ o_rgba,
    int32_t* __restrict__ o_frames, int32_t* __restrict__ o_cams, float* __restrict__ o_minmax, int64_t* __restrict__ o_pixel,
    float* __restrict__ out_t, int64_t* __restrict__ out_ray)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= num_rays) return;
    const int32_t r = order[i];
    const int32_t src = ray_start[r], n = ray_cnt[r], dst = out_offset[i];
    for (int32_t j = lane; j < n; j += 64) {
        out_t[dst + j] = t_stage[src + j];
        out_ray[dst + j] = i;
    }
    if (lane < 3) { o_origins[i * 3 + lane] = origins[(int64_t)r * 3 + lane]; o_dirs[i * 3 + lane] = dirs[(int64_t)r * 3 + lane]; }
    else if (lane < 7) o_rgba[i * 4 + lane - 3] = rgba[(int64_t)r * 4 + lane - 3];
    else if (lane < 9) o_minmax[i * 2 + lane - 7] = minmax[(int64_t)r * 2 + lane - 7];
    else if (lane == 9) { o_frames[i] = frames[r]; o_cams[i] = cams[r]; }
    else if (lane == 10 && pixel) o_pixel[i] = pixel[r];
}

extern "C" int hrf_pack_runs_sorted(const int32_t* order, const int32_t* ray_start, const int32_t* ray_cnt,
                                    const int32_t* out_offset_sorted, const float* t_stage, int64_t num_rays,
                                    const float* origins, const float* dirs, const float* rgba, const int32_t* frames,
                                    const int32_t* cams, const float* minmax, const int64_t* pixel, float* o_origins,
                                    float* o_dirs, float* o_rgba, int32_t* o_frames, int32_t* o_cams, float* o_minmax,
                                    int64_t* o_pixel, float* out_t, int64_t* out_ray, hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(order && ray_start && ray_cnt && out_offset_sorted && t_stage && out_t && out_ray, "NULL argument");
    HRF_CHECK_ARG(origins && dirs && rgba && frames && cams && minmax, "NULL ray input");
    HRF_CHECK_ARG(o_origins && o_dirs && o_rgba && o_frames && o_cams && o_minmax && (!pixel || o_pixel), "NULL ray output");
    hipLaunchKernelGGL(k_pack_runs_sorted, dim3(hrf_blocks(num_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream, order,
                       ray_start, ray_cnt, out_offset_sorted, t_stage, num_rays, origins, dirs, rgba, frames, cams, minmax,
                       pixel, o_origins, o_dirs, o_rgba, o_frames, o_cams, o_minmax, o_pixel, out_t, out_ray);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The batch-growing loop of Trainer.train (humanrf/trainer.py:138-163) replayed on the device over rays that have
// ALREADY been marched speculatively. The reference draws rays_initial rays, prunes, and keeps drawing
// int((samples_max - total_samples) / (total_samples / total_rays)) more until total_samples >= 0.9 * samples_max;
// every iteration costs it a sampler call, a prune pass and several host synchronisations. Here the drawn rays of a
// step are a prefix of a set whose sampler stages ran ahead of time, per-ray results do not depend on what else is in
// a launch, and the survivors of ANY drawn prefix are prefix sums that exist after one march launch:
//   compacted rays among the first x drawn rays   = slot[x]                      (exclusive scan of the ray mask)
//   visible samples of the first k compacted rays = out_off[k - ray_base]        (exclusive scan of the per-ray counts)
// so the loop's decisions (same double-precision arithmetic as the Python statements) need no further launches or
// read-backs. One thread; the loop runs 1-3 times.
// plan (int64[16]) out: [0] done, [1] iterations run, [2] drawn rays used, [3] next r0, [4] compacted rays up to `used` (absolute),
//   [5] visible samples of this chunk's rays, [6] error (1: zero samples per ray, the reference's assert), [7] total drawn rays,
//   [8] *extra, [9] slot[spec_end], [10..12] visible-sample offsets of the chunk's compacted rays at rays * 1/4, 2/4, 3/4,
//   [13] rays of the chunk, [14], [15] zero.
// ------------------------------------------------------------------------------------------------
__global__ void k_batch_plan(const int32_t* __restrict__ slot, const int32_t* __restrict__ out_off, int64_t ray_base,
                             int64_t used, int64_t spec_end, int64_t r0, int64_t total_rays, int64_t total_samples,
                             int64_t samples_max, const int32_t* __restrict__ extra, int64_t* __restrict__ plan)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t samples_before = total_samples;
    int64_t iters = 0, done = 0, err = 0;
    int64_t chunk_samples = 0;
    while (used + r0 <= spec_end) {
        used += r0;                                                   // trainer.py:144 (next(loader) with batch_size r0)
        total_rays += r0;                                             // trainer.py:154
        chunk_samples = (int64_t)out_off[slot[used] - ray_base];
        total_samples = samples_before + chunk_samples;               // trainer.py:155
        ++iters;
        if ((double)total_samples < 0.9 * (double)samples_max) {      // trainer.py:156
            const double avg = (double)total_samples / (double)total_rays;   // trainer.py:157
            if (!(avg > 0.0)) { err = 1; break; }                     // trainer.py:158
            r0 = (int64_t)((double)(samples_max - total_samples) / avg);     // trainer.py:160-161 (int() truncates)
        } else {
            done = 1;
            break;
        }
    }
    plan[0] = done; plan[1] = iters; plan[2] = used; plan[3] = r0; plan[4] = (int64_t)slot[used]; plan[5] = chunk_samples;
    plan[6] = err; plan[7] = total_rays;
    plan[8] = extra ? (int64_t)*extra : 0;   // one more device scalar the caller wants in the same read-back
    plan[9] = (int64_t)slot[spec_end];       // compacted rays among all the drawn rays marched so far
    // sample offsets at the quarter points of this chunk's rays (relative to the chunk): where a caller may cut the batch
    // into pieces that end on ray boundaries (the trainer pipelines the pieces over two streams)
    const int64_t rays_c = (int64_t)slot[used] - ray_base;
    for (int k = 1; k <= 3; ++k) plan[9 + k] = (int64_t)out_off[rays_c * k / 4];
    plan[13] = rays_c; plan[14] = 0; plan[15] = 0;
}

extern "C" int hrf_batch_plan(const int32_t* slot, const int32_t* out_offset, int64_t ray_base, int64_t used,
                              int64_t spec_end, int64_t r0, int64_t total_rays, int64_t total_samples,
                              int64_t samples_max, const int32_t* extra, int64_t* plan, hrf_stream_t stream)
{
    HRF_CHECK_ARG(slot && out_offset && plan, "NULL argument");
    HRF_CHECK_ARG(used >= 0 && spec_end >= used && r0 > 0 && samples_max > 0 && ray_base >= 0, "bad loop state");
    hipLaunchKernelGGL(k_batch_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, slot, out_offset, ray_base, used, spec_end,
                       r0, total_rays, total_samples, samples_max, extra, plan);
    HRF_CHECK_LAUNCH();
    return 0;
}
