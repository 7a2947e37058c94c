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

#define MARCH_ROW 40  // halves per LDS feature row (32 + 8 pad: 80-byte rows, 8-byte aligned fragments)

// CH = samples per march step (16, 32 or 64). The 64 lanes of the wavefront always work on CH consecutive samples
// of ONE ray: lane = (sample = lane % CH, part = lane / CH) and a lane encodes the 16*CH/64 levels
// {part * LPL + k}. Small CH wastes fewer samples past the point where the ray saturates (the march can only stop
// at a step boundary) at the price of more steps per ray; the encode work per step is the same 64-lane-wide gather.
template <int CH>
__global__ __launch_bounds__(128, 4) void k_prune_march(
    const float* __restrict__ ray_o, const float* __restrict__ ray_d, const int32_t* __restrict__ ray_frames,
    const int32_t* __restrict__ ray_start, const float* __restrict__ t0, const float* __restrict__ jitter, float step,
    float eps, float thre, const int32_t* __restrict__ f2s, const float* __restrict__ f2l,
    const __half2* __restrict__ tables, const float* __restrict__ vectors, const hrf_segment_meta* __restrict__ segs,
    int vec_res, const _Float16* __restrict__ w1, const _Float16* __restrict__ w2, float density_scale, int64_t num_rays,
    const int32_t* __restrict__ num_rays_dev, int64_t capacity, float* __restrict__ t_stage,
    float* __restrict__ sigma_stage, int32_t* __restrict__ ray_cnt, int32_t* __restrict__ ray_evaluated)
{
    constexpr int LPL = 16 * CH / 64;  // levels per lane
    constexpr int TILES = CH / 16;     // MFMA column tiles per step
    __shared__ __attribute__((aligned(16))) _Float16 s_w1[64 * (32 + WPAD)];
    __shared__ __attribute__((aligned(16))) _Float16 s_w2[16 * (64 + WPAD)];
    __shared__ __attribute__((aligned(16))) _Float16 s_feat[2][CH * MARCH_ROW];
    // Two wavefronts per workgroup: rays differ a lot in length, so small workgroups let the dispatcher balance;
    // two (not one) keeps the LDS footprint per wavefront low enough for 4 wavefronts per SIMD.
    stage_rm(s_w1, w1, 64, 32);
    stage_rm(s_w2, w2, 16, 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int smp = lane % CH, part = lane / CH;
    _Float16* feat = s_feat[threadIdx.x >> 6];
    const int64_t wave_id = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 2;

    const int64_t live_rays = num_rays_dev ? min((int64_t)*num_rays_dev, num_rays) : num_rays;
    for (int64_t r = wave_id; r < num_rays; r += n_waves) {
        if (r >= live_rays) {  // host passed an upper bound: slots beyond the device-side count hold nothing
            if (lane == 0) { ray_cnt[r] = 0; if (ray_evaluated) ray_evaluated[r] = 0; }
            continue;
        }
        const int32_t rb = ray_start[r], re = ray_start[r + 1];
        const float ox = ray_o[r * 3 + 0], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
        const float dx = ray_d[r * 3 + 0], dy = ray_d[r * 3 + 1], dz = ray_d[r * 3 + 2];
        const int32_t frame = ray_frames[r];
        const int seg = __builtin_amdgcn_readfirstlane(f2s[frame]);
        const float tl = f2l[frame];
        const hrf_segment_meta* sm = segs + seg;
        const __half2* tbase = tables + sm->table_offset;
        const uint32_t entries = sm->entries;
        const float* vbase = vectors + (size_t)seg * 4 * vec_res * ENC_F;
        float T = 1.0f;
        int32_t kept = 0, evaluated = 0;
        for (int32_t cb = rb; cb < re; cb += CH) {
            const int32_t i = cb + smp;
            const bool valid = i < re && i < capacity;  // capacity: size of t0 / jitter / t_stage
            float t = 0.0f;
            if (valid) {
                t = t0[i];
                if (jitter) t = t + jitter[i] * step;  // volume_rendering.py:63-64
            }
            EncCoords q;
            q.c[0] = (ox + t * dx) + 0.5f;  // volume_rendering.py:68-69, humanrf.py:175
            q.c[1] = (oy + t * dy) + 0.5f;
            q.c[2] = (oz + t * dz) + 0.5f;
            q.c[3] = tl;
            int vc0[4], vc1[4];
            float vfr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) hrf_vec_tap(q.c[k], vec_res, vc0[k], vc1[k], vfr[k]);
#pragma unroll 1
            for (int li = 0; li < LPL; ++li) {
                const int l = part * LPL + li;
                const hrf_level_meta lv = sm->levels[l];
                float fe[4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a, b, cc;
                    enc_pick(q, e, a, b, cc);
                    const __half2* tb = tbase + (size_t)e * entries + lv.offset;
                    float f0, f1;
                    enc_gather(tb, a, b, cc, lv, f0, f1);
                    const float2 hf = __half22float2(__floats2half2_rn(f0, f1));
                    fe[e][0] = hf.x; fe[e][1] = hf.y;
                }
                float sv[4][2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 v0 = *(const float2*)(vbase + ((size_t)k * vec_res + vc0[k]) * ENC_F + 2 * l);
                    const float2 v1 = *(const float2*)(vbase + ((size_t)k * vec_res + vc1[k]) * ENC_F + 2 * l);
                    sv[k][0] = v0.x + vfr[k] * (v1.x - v0.x);
                    sv[k][1] = v0.y + vfr[k] * (v1.y - v0.y);
                }
                const float r0 = ((fe[0][0] * sv[3][0] + fe[1][0] * sv[2][0]) + fe[2][0] * sv[0][0]) + fe[3][0] * sv[1][0];
                const float r1 = ((fe[0][1] * sv[3][1] + fe[1][1] * sv[2][1]) + fe[2][1] * sv[0][1]) + fe[3][1] * sv[1][1];
                *(__half2*)(feat + smp * MARCH_ROW + 2 * l) = __floats2half2_rn(r0, r1);
            }
            // wavefront-private LDS: no barrier, only ordering of this wavefront's own DS operations
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float sigma = 0.0f;
            // weight fragments are re-read from LDS per step (8-byte reads) instead of living in 24 VGPRs across
            // the gather phase: keeps the kernel at 4 wavefronts per SIMD
            h4 a1[4][2], a2[4];
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                a1[ht][0] = afrag(s_w1, 32, ht, 0, lane);
                a1[ht][1] = afrag(s_w1, 32, ht, 1, lane);
                a2[ht] = afrag(s_w2, 64, 0, ht, lane);
            }
#pragma unroll
            for (int tt = 0; tt < TILES; ++tt) {
                const h4 x0 = *(const h4*)(feat + (16 * tt + c) * MARCH_ROW + 4 * g);
                const h4 x1 = *(const h4*)(feat + (16 * tt + c) * MARCH_ROW + 16 + 4 * g);
                f4 o = f4zero();
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) {
                    f4 acc = mfma16(a1[ht][0], x0, f4zero());
                    acc = mfma16(a1[ht][1], x1, acc);
                    o = mfma16(a2[ht], relu_h4(acc), o);
                }
                // h0 of sample 16*tt + c sits in lane c (g == 0): hand it to every lane whose sample that is
                const float h0 = hround(o[0]);
                const float mine = __shfl(h0, smp & 15, 64);
                if ((smp >> 4) == tt) sigma = expf(mine) * density_scale;  // truncated_exp forward, humanrf.py:184
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // visibility (same sequential product as k_visibility / oracle orc_visibility); lanes with part > 0
            // duplicate the sample of lane smp and take no part in the compaction
            const float a = valid ? (1.0f - expf(-sigma * step)) : 0.0f;  // volume_rendering.py:76
            const float om = 1.0f - a;
            float myT = 0.0f;
            const int cnt = min(CH, re - cb);
            for (int k = 0; k < cnt; ++k) {
                if (smp == k) myT = T;
                T = T * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, om), k));
            }
            const bool vis = valid && part == 0 && (myT >= eps) && (a >= thre);
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
                               hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(ray_origins && ray_dirs && ray_frames && ray_start && t0, "NULL ray / sample input");
    HRF_CHECK_ARG(frame_to_segment && frame_to_local && tables && vectors && segments && w1 && w2, "NULL model input");
    HRF_CHECK_ARG(t_stage && ray_cnt, "NULL output");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1, "bad segment count / vector resolution");
    unsigned blocks = (unsigned)std::min<int64_t>((num_rays + 1) / 2, 1 << 20);
    int ch = 64;  // samples per march step (measured: 16 / 32 save <15% of the encoded samples -- most non-visible
                  // samples are transparent ones in front of the surface, not step padding -- and cost more per step)
    if (const char* dbg = getenv("HRF_MARCH_CHUNK")) ch = atoi(dbg);  // tuning aid: 16 / 32 / 64
#define HRF_LAUNCH_MARCH(CHV)                                                                                         \
    hipLaunchKernelGGL(k_prune_march<CHV>, dim3(blocks), dim3(128), 0, (hipStream_t)stream, ray_origins, ray_dirs,     \
                       ray_frames, ray_start, t0, jitter, step, early_stop_eps, alpha_thre, frame_to_segment,          \
                       frame_to_local, (const __half2*)tables, vectors, segments, vec_res, (const _Float16*)w1,       \
                       (const _Float16*)w2, density_scale, num_rays, num_rays_dev, capacity, t_stage, sigma_stage,     \
                       ray_cnt, ray_evaluated)
    if (ch == 64) HRF_LAUNCH_MARCH(64); else if (ch == 32) HRF_LAUNCH_MARCH(32); else HRF_LAUNCH_MARCH(16);
#undef HRF_LAUNCH_MARCH
    HRF_CHECK_LAUNCH();
    return 0;
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
