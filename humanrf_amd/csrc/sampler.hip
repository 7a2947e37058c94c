// Occupancy-grid ray sampler for gfx950: replaces actorshq/dataset/native/ray_sampler.cu and
// occupancy_grid.cu. Arithmetic is the bit-defined form of oracle/sampler_oracle.c (fp32, no FMA
// contraction, integer texture predicate); structure is MI355X-native: plain HBM uint8 volumes instead of
// texture objects, one wavefront per ray with ballot/prefix-popcount compaction for the sample stage, and
// no host synchronisation inside (the reference has >= 4 implicit syncs, ray_sampler.cu:256-323).
#include "hrf_common.h"
#include <algorithm>
#include <vector>

// ------------------------------------------------------------------------------------------------
// Occupancy ring (class OccupanyGrid, occupancy_grid.cu:8-88)
// ------------------------------------------------------------------------------------------------
struct HrfOccRing {
    uint64_t res;
    int slots;
    int next;
    uint8_t* base;  // slots * (res^3 + mip) bytes
    size_t slot_bytes;
};

// Size of the coarse mip that follows every volume (0 when the resolution is not a multiple of HRF_MIP).
static inline size_t hrf_mip_dim(uint64_t G) { return (G % HRF_MIP == 0) ? (size_t)(G / HRF_MIP) : 0; }
// ... and of the block mip behind it (one byte per 16^3 texels, set when a texel within the block widened by >= 4 texels on every
// side is non-zero): what lets a ray that cannot see any occupied texel leave the occupancy march at once (k_sampler_rays_coop).
#define HRF_MIP2 16
static inline size_t hrf_mip2_dim(uint64_t G) { return (G % HRF_MIP2 == 0) ? (size_t)(G / HRF_MIP2) : 0; }

__global__ __launch_bounds__(256) void k_build_mip(const uint8_t* __restrict__ g, int G, int C, uint8_t* __restrict__ mip)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * C * C) return;
    const int cx = (int)(i % C), cy = (int)((i / C) % C), cz = (int)(i / ((int64_t)C * C));
    unsigned any = 0;
    for (int z = max(HRF_MIP * cz - 1, 0); z <= min(HRF_MIP * cz + HRF_MIP, G - 1); ++z)
        for (int y = max(HRF_MIP * cy - 1, 0); y <= min(HRF_MIP * cy + HRF_MIP, G - 1); ++y)
            for (int x = max(HRF_MIP * cx - 1, 0); x <= min(HRF_MIP * cx + HRF_MIP, G - 1); ++x)
                any |= g[((size_t)z * G + y) * G + x];
    mip[i] = any ? 1 : 0;
}

// Block mip from the 4^3 mip: mip cell c stands for texels [4c - 1, 4c + 4], so the cells [4b - 1, 4b + 4] of block b cover texels
// [16b - 5, 16b + 20]: the block's 16 texels widened by 5 below and above.
__global__ __launch_bounds__(256) void k_build_mip2(const uint8_t* __restrict__ mip, int C, int C2, uint8_t* __restrict__ mip2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C2 * C2 * C2) return;
    const int bx = (int)(i % C2), by = (int)((i / C2) % C2), bz = (int)(i / ((int64_t)C2 * C2));
    const int R = HRF_MIP2 / HRF_MIP;
    unsigned any = 0;
    for (int z = max(R * bz - 1, 0); z <= min(R * bz + R, C - 1); ++z)
        for (int y = max(R * by - 1, 0); y <= min(R * by + R, C - 1); ++y)
            for (int x = max(R * bx - 1, 0); x <= min(R * bx + R, C - 1); ++x)
                any |= mip[((size_t)z * C + y) * C + x];
    mip2[i] = any ? 1 : 0;
}

extern "C" int hrf_occgrid_create(uint64_t grid_resolution, int buffer_size, void** out_handle)
{
    HRF_CHECK_ARG(out_handle != nullptr, "out_handle is NULL");
    HRF_CHECK_ARG(grid_resolution > 0 && grid_resolution <= 4096, "grid_resolution out of range");
    HRF_CHECK_ARG(buffer_size > 0, "buffer_size must be positive");
    HrfOccRing* r = new HrfOccRing{grid_resolution, buffer_size, 0, nullptr, 0};
    const size_t C = hrf_mip_dim(grid_resolution), C2 = hrf_mip2_dim(grid_resolution);
    r->slot_bytes = ((size_t)grid_resolution * grid_resolution * grid_resolution + C * C * C + C2 * C2 * C2 + 255) / 256 * 256;
    size_t bytes = (size_t)buffer_size * r->slot_bytes;
    hipError_t e = hipMalloc((void**)&r->base, bytes);
    if (e != hipSuccess) {
        delete r;
        hrf_set_error("hrf_occgrid_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return 2;
    }
    *out_handle = r;
    return 0;
}

extern "C" int hrf_occgrid_add(void* handle, const uint8_t* grid, uint64_t g0, uint64_t g1, uint64_t g2,
                               hrf_stream_t stream, int64_t* out_texture_host)
{
    HRF_CHECK_ARG(handle != nullptr && grid != nullptr && out_texture_host != nullptr, "NULL argument");
    HrfOccRing* r = (HrfOccRing*)handle;
    // occupancy_grid.cu:60-63
    HRF_CHECK_ARG(g0 == r->res && g1 == r->res && g2 == r->res, "Provided grid doesn't have the correct resolution!");
    int used = r->next;
    r->next = (r->next + 1) % r->slots;  // ring-slot reuse, occupancy_grid.cu:65-66
    size_t bytes = (size_t)r->res * r->res * r->res;
    uint8_t* dst = r->base + (size_t)used * r->slot_bytes;
    hipError_t e = hipMemcpyAsync(dst, grid, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) {
        hrf_set_error("hrf_occgrid_add: copy failed: %s", hipGetErrorString(e));
        return 2;
    }
    const size_t C = hrf_mip_dim(r->res);
    if (C > 0) {
        hipLaunchKernelGGL(k_build_mip, dim3(hrf_blocks((int64_t)(C * C * C), 256)), dim3(256), 0, (hipStream_t)stream,
                           dst, (int)r->res, (int)C, dst + bytes);
        const size_t C2 = hrf_mip2_dim(r->res);
        if (C2 > 0)
            hipLaunchKernelGGL(k_build_mip2, dim3(hrf_blocks((int64_t)(C2 * C2 * C2), 256)), dim3(256), 0, (hipStream_t)stream,
                               dst + bytes, (int)C, (int)C2, dst + bytes + C * C * C);
        HRF_CHECK_LAUNCH();
    }
    *out_texture_host = (int64_t)(uintptr_t)dst;
    return 0;
}

extern "C" int hrf_occgrid_destroy(void* handle)
{
    if (!handle) return 0;
    HrfOccRing* r = (HrfOccRing*)handle;
    (void)hipFree(r->base);
    delete r;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Stage 1: pixel -> ray, slab test, occupancy march (ray_sampler.cu:11-147)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gmin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float gmax(float a, float b) { return (a < b) ? b : a; }

// AABB-only modes (get_{rays,samples}_aabb_minmax); the occupancy modes use the cooperative kernel below.
__global__ __launch_bounds__(256) void k_sampler_rays(
    const float* __restrict__ inverse_krs, const float* __restrict__ camera_origins,
    const uint8_t* __restrict__ landscape, const int64_t* __restrict__ ray_indices,
    const int64_t* __restrict__ grid_textures, const float* __restrict__ aabb,
    const uint8_t* __restrict__ light_mask, int64_t num_rays, int G, int width_in, int height_in, float step,
    float* __restrict__ out_dirs, float* __restrict__ out_minmax, uint8_t* __restrict__ out_mask,
    int32_t* __restrict__ out_count)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    int width = width_in, height = height_in;
    const int64_t idx = ray_indices[r];
    const int image = (int)(idx / ((int64_t)width * height));
    if (!landscape[image]) { int t = width; width = height; height = t; }
    const float px = (float)(idx % width) + 0.5f;
    const float py = (float)((idx / width) % height) + 0.5f;
    const float* m = inverse_krs + (size_t)image * 9;
    const float ox = camera_origins[image * 3 + 0], oy = camera_origins[image * 3 + 1], oz = camera_origins[image * 3 + 2];
    float vx = (m[0] * px + m[3] * py) + m[6] * 1.0f;
    float vy = (m[1] * px + m[4] * py) + m[7] * 1.0f;
    float vz = (m[2] * px + m[5] * py) + m[8] * 1.0f;
    float dot = (vx * vx + vy * vy) + vz * vz;
    float inv = 1.0f / sqrtf(dot);
    const float dx = vx * inv, dy = vy * inv, dz = vz * inv;

    // compute_aabb_minmax (ray_sampler.cu:11-26)
    float mnx, mny, mnz, mxx, mxy, mxz;
    {
        float i0 = 1.0f / dx, i1 = 1.0f / dy, i2 = 1.0f / dz;
        float a0 = (aabb[0] - ox) * i0, b0 = (aabb[3] - ox) * i0;
        float a1 = (aabb[1] - oy) * i1, b1 = (aabb[4] - oy) * i1;
        float a2 = (aabb[2] - oz) * i2, b2 = (aabb[5] - oz) * i2;
        mnx = gmin(a0, b0); mxx = gmax(a0, b0);
        mny = gmin(a1, b1); mxy = gmax(a1, b1);
        mnz = gmin(a2, b2); mxz = gmax(a2, b2);
    }
    float tmin = gmax(mnx, gmax(mny, mnz));
    float tmax = gmin(mxx, gmin(mxy, mxz));

    bool mask = tmin < tmax;
    if (light_mask) mask = mask && !light_mask[idx];  // ray_sampler.cu:254-257
    out_dirs[r * 3 + 0] = dx; out_dirs[r * 3 + 1] = dy; out_dirs[r * 3 + 2] = dz;
    out_minmax[r * 2 + 0] = tmin; out_minmax[r * 2 + 1] = tmax;
    out_mask[r] = mask ? 1 : 0;
    out_count[r] = mask ? (int32_t)((tmax - tmin) / step) : 0;  // ray_sampler.cu:283-285
}

// Pre-pass of the occupancy march: pixel -> ray and box segment as below, then the conservative "can any march position see an
// occupied texel?" test on the 16^3-block mip (see k_sampler_rays_coop for the argument). A ray that cannot gets its outputs here
// (mask 0, count 0, direction); the others are appended to `list` for the exact march. 16 lanes per ray; a workgroup walks its
// share of the rays (16 per round), collects its survivors in LDS and reserves their place in the list with ONE returning atomic
// (one per wavefront was the first version: 60 000 returning atomics on one address, 0.24 ms for a kernel of 0.02).
#define PRE_CAP 4096    // survivors a workgroup can hold (the launcher sizes the grid so that a workgroup sees at most this many rays)
__global__ __launch_bounds__(256) void k_sampler_prepass(
    const float* __restrict__ inverse_krs, const float* __restrict__ camera_origins,
    const uint8_t* __restrict__ landscape, const int64_t* __restrict__ ray_indices,
    const int64_t* __restrict__ grid_textures, const float* __restrict__ aabb, int64_t num_rays, int G, int width_in, int height_in,
    float* __restrict__ out_dirs, float* __restrict__ out_minmax, uint8_t* __restrict__ out_mask, int32_t* __restrict__ out_count,
    int32_t* __restrict__ list, int32_t* __restrict__ list_n)
{
    __shared__ int32_t s_list[PRE_CAP];
    __shared__ int32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane % 16;
    const int C = G / HRF_MIP, C2 = G / HRF_MIP2;          // (the launcher only takes this path when G is a multiple of 16)
    const float dt = 4.0f / (float)G, to_block = (float)G * (1.0f / (float)HRF_MIP2);
    for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < num_rays; r0 += (int64_t)gridDim.x * 16) {
        const int64_t r_raw = r0 + threadIdx.x / 16;
        const bool live = r_raw < num_rays;
        const int64_t r = live ? r_raw : num_rays - 1;
        int width = width_in, height = height_in;
        const int64_t idx = ray_indices[r];
        const int image = (int)(idx / ((int64_t)width * height));
        if (!landscape[image]) { int t = width; width = height; height = t; }
        const float px = (float)(idx % width) + 0.5f;
        const float py = (float)((idx / width) % height) + 0.5f;
        const float* m = inverse_krs + (size_t)image * 9;
        const float ox = camera_origins[image * 3 + 0], oy = camera_origins[image * 3 + 1], oz = camera_origins[image * 3 + 2];
        float vx = (m[0] * px + m[3] * py) + m[6] * 1.0f;
        float vy = (m[1] * px + m[4] * py) + m[7] * 1.0f;
        float vz = (m[2] * px + m[5] * py) + m[8] * 1.0f;
        float dot = (vx * vx + vy * vy) + vz * vz;
        float inv = 1.0f / sqrtf(dot);
        const float dx = vx * inv, dy = vy * inv, dz = vz * inv;
        float tmin, tmax;
        {
            float i0 = 1.0f / dx, i1 = 1.0f / dy, i2 = 1.0f / dz;
            float a0 = (aabb[0] - ox) * i0, b0 = (aabb[3] - ox) * i0;
            float a1 = (aabb[1] - oy) * i1, b1 = (aabb[4] - oy) * i1;
            float a2 = (aabb[2] - oz) * i2, b2 = (aabb[5] - oz) * i2;
            tmin = gmax(gmin(a0, b0), gmax(gmin(a1, b1), gmin(a2, b2)));
            tmax = gmin(gmax(a0, b0), gmin(gmax(a1, b1), gmax(a2, b2)));
        }
        hrf_gbytes mip2 = (hrf_gbytes)(uintptr_t)grid_textures[image] + (size_t)G * G * G + (size_t)C * C * C;
        const float span = tmax - tmin;
        bool seen = false;
        if (!(span < 4.0f)) seen = true;      // (not a finite segment of the unit box: leave it to the exact march)
        else if (span > 0.0f) {
            for (float a = (float)j * dt; a < span + dt; a += 16.0f * dt) {
                const float t = tmin + fminf(a, span);
                const float px3 = (ox + dx * t) + 0.5f, py3 = (oy + dy * t) + 0.5f, pz3 = (oz + dz * t) + 0.5f;
                const int bx = (int)fminf(fmaxf(px3 * to_block, 0.0f), (float)(C2 - 1));
                const int by = (int)fminf(fmaxf(py3 * to_block, 0.0f), (float)(C2 - 1));
                const int bz = (int)fminf(fmaxf(pz3 * to_block, 0.0f), (float)(C2 - 1));
                seen |= mip2[((size_t)bz * C2 + by) * C2 + bx] != 0;
            }
        }
        const unsigned long long bal = __ballot(seen && live);
        const bool keep = ((bal >> (16 * (lane / 16))) & 0xFFFFull) != 0ull;       // this lane's ray goes on to the exact march
        if (live && j == 0) {
            if (keep) s_list[atomicAdd(&s_n, 1)] = (int32_t)r;
            else {
                out_dirs[r * 3 + 0] = dx; out_dirs[r * 3 + 1] = dy; out_dirs[r * 3 + 2] = dz;
                out_minmax[r * 2 + 0] = tmax; out_minmax[r * 2 + 1] = tmax;      // (tmin >= tmax: an empty range; unspecified by the ABI)
                out_mask[r] = 0;
                out_count[r] = 0;
            }
        }
    }
    __syncthreads();
    const int n_here = s_n;
    if (threadIdx.x == 0) s_base = n_here ? atomicAdd(list_n, n_here) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_here; i += blockDim.x) list[s_base + i] = s_list[i];
}

// Cooperative variant of the occupancy march: 16 lanes per ray test 16 consecutive march steps at once.
// The reference's loop is `while (t < max) { if (occupied(t)) break; t += step; }` with t accumulated by repeated
// fp32 addition; lane j reproduces exactly the value after j more additions (j <= 15 sequential adds), the group
// then takes the FIRST lane whose step would have ended the loop (hit, or t no longer inside), so tmin / tmax are
// bit-identical to the sequential march while the dependent chain of texel loads shrinks 16-fold.
#define COOP 16
__device__ __forceinline__ unsigned coop_ballot(bool p, int lane)
{
    return (unsigned)((__ballot(p) >> (COOP * (lane / COOP))) & 0xFFFFu);
}

__global__ __launch_bounds__(256) void k_sampler_rays_coop(
    const float* __restrict__ inverse_krs, const float* __restrict__ camera_origins,
    const uint8_t* __restrict__ landscape, const int64_t* __restrict__ ray_indices,
    const int64_t* __restrict__ grid_textures, const float* __restrict__ aabb,
    const uint8_t* __restrict__ light_mask, int64_t num_rays, int G, int width_in, int height_in, float step,
    float* __restrict__ out_dirs, float* __restrict__ out_minmax, uint8_t* __restrict__ out_mask,
    int32_t* __restrict__ out_count, const int32_t* __restrict__ list, const int32_t* __restrict__ list_n)
{
    const int lane = threadIdx.x & 63, j = lane % COOP;
    // list != NULL: the rays that passed k_sampler_prepass (their ids in any order, *list_n of them): a wavefront's four rays march
    // until the last of them is done, so rays that can leave early only pay off when they never enter a wavefront
    const int64_t slot = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / COOP;
    const int64_t n_slots = list ? (int64_t)*list_n : num_rays;
    if (((int64_t)blockIdx.x * blockDim.x) / COOP >= n_slots) return;          // (whole workgroup beyond the list)
    const bool live = slot < n_slots;
    const int64_t r_raw = list ? (int64_t)list[live ? slot : n_slots - 1] : slot;
    const int64_t r = live ? r_raw : (list ? r_raw : num_rays - 1);  // idle groups shadow the last ray (no divergent exit before ballots)
    int width = width_in, height = height_in;
    const int64_t idx = ray_indices[r];
    const int image = (int)(idx / ((int64_t)width * height));
    if (!landscape[image]) { int t = width; width = height; height = t; }
    const float px = (float)(idx % width) + 0.5f;
    const float py = (float)((idx / width) % height) + 0.5f;
    const float* m = inverse_krs + (size_t)image * 9;
    const float ox = camera_origins[image * 3 + 0], oy = camera_origins[image * 3 + 1], oz = camera_origins[image * 3 + 2];
    float vx = (m[0] * px + m[3] * py) + m[6] * 1.0f;
    float vy = (m[1] * px + m[4] * py) + m[7] * 1.0f;
    float vz = (m[2] * px + m[5] * py) + m[8] * 1.0f;
    float dot = (vx * vx + vy * vy) + vz * vz;
    float inv = 1.0f / sqrtf(dot);
    const float dx = vx * inv, dy = vy * inv, dz = vz * inv;
    float mnx, mny, mnz, mxx, mxy, mxz;
    {
        float i0 = 1.0f / dx, i1 = 1.0f / dy, i2 = 1.0f / dz;
        float a0 = (aabb[0] - ox) * i0, b0 = (aabb[3] - ox) * i0;
        float a1 = (aabb[1] - oy) * i1, b1 = (aabb[4] - oy) * i1;
        float a2 = (aabb[2] - oz) * i2, b2 = (aabb[5] - oz) * i2;
        mnx = gmin(a0, b0); mxx = gmax(a0, b0);
        mny = gmin(a1, b1); mxy = gmax(a1, b1);
        mnz = gmin(a2, b2); mxz = gmax(a2, b2);
    }
    float tmin = gmax(mnx, gmax(mny, mnz));
    float tmax = gmin(mxx, gmin(mxy, mxz));
    hrf_gbytes g = (hrf_gbytes)(uintptr_t)grid_textures[image];
    const int C = (G % HRF_MIP == 0) ? G / HRF_MIP : 0;
    hrf_gbytes mip = C ? g + (size_t)G * G * G : nullptr;
    const float mstep = 0.5f / (float)G;
    const float aabb_max = tmax;
    const int gbase = lane - j;  // first lane of this ray's group

    // Can ANY position of the march see an occupied texel? Nine in ten drawn rays cannot (they pass the body by) and would walk
    // the whole box -- ~2 G steps -- to find that out. The group samples the box segment every 4 texels of arc length and looks the
    // samples up in the 16^3-block mip: a march position is within 2 texels of a sample and its trilinear tap within 1.5 texels
    // of the position, and a block is set when a texel within >= 4.5 texels of it (5 below, 4 above its cells) is non-zero, so
    // "no sample in a set block" implies that every predicate of the forward march is false: the exact march would end at the box
    // exit with tmin >= tmax and mask 0, which is what such a ray gets here at once. (For a ray with mask 0 only mask and count are
    // defined outputs: every consumer compacts by the mask first.) Rays that pass the test take the exact march below.
    bool maybe = true;
    {
        const int C2 = (C && G % HRF_MIP2 == 0 && !list) ? G / HRF_MIP2 : 0;      // (a listed ray has passed this test already)
        if (C2) {
            hrf_gbytes mip2 = mip + (size_t)C * C * C;
            const float dt = 4.0f / (float)G, span = tmax - tmin, to_block = (float)G * (1.0f / (float)HRF_MIP2);
            bool seen = false;
            if (!(span < 4.0f)) seen = true;      // (not a finite segment of the unit box: leave it to the exact march)
            else if (span > 0.0f) {
                for (float a = (float)j * dt; a < span + dt; a += (float)COOP * dt) {
                    const float t = tmin + fminf(a, span);
                    const float px3 = (ox + dx * t) + 0.5f, py3 = (oy + dy * t) + 0.5f, pz3 = (oz + dz * t) + 0.5f;
                    const int bx = (int)fminf(fmaxf(px3 * to_block, 0.0f), (float)(C2 - 1));
                    const int by = (int)fminf(fmaxf(py3 * to_block, 0.0f), (float)(C2 - 1));
                    const int bz = (int)fminf(fmaxf(pz3 * to_block, 0.0f), (float)(C2 - 1));
                    seen |= mip2[((size_t)bz * C2 + by) * C2 + bx] != 0;
                }
            }
            maybe = coop_ballot(seen, lane) != 0u;
            if (!maybe) tmin = aabb_max;       // (tmin >= tmax: mask 0)
        }
    }

    // forward march (ray_sampler.cu:36-45)
    {
        float tcur = tmin;
        bool done = !maybe;
        while (__any(!done)) {
            float tj = tcur;
            for (int i = 0; i < j; ++i) tj += mstep;
            const bool inside = tj < aabb_max;
            const bool stop = !done && (!inside || hrf_occ_at(g, mip, G, C, ox, oy, oz, dx, dy, dz, tj));
            const unsigned b = coop_ballot(stop, lane);
            if (!done) {
                if (b) {
                    const int first = __ffs((int)b) - 1;
                    tmin = __shfl(tj, gbase + first, 64);
                    done = true;
                } else {
                    tcur = __shfl(tj, gbase + COOP - 1, 64) + mstep;
                }
            }
        }
    }
    if (tmin < aabb_max) {  // refine (ray_sampler.cu:47-64); identical in every lane of the group
        float refine = -mstep * 0.5f;
        for (int i = 0; i < 5; ++i) {
            tmin += refine;
            if (hrf_occ_at(g, mip, G, C, ox, oy, oz, dx, dy, dz, tmin)) refine = -fabsf(refine) * 0.5f;
            else refine = fabsf(refine) * 0.5f;
        }
    }
    // backward march from the box exit (ray_sampler.cu:66-75)
    {
        float tcur = tmax;
        bool done = !maybe;
        while (__any(!done)) {
            float tj = tcur;
            for (int i = 0; i < j; ++i) tj -= mstep;
            const bool inside = tj > tmin;
            const bool stop = !done && (!inside || hrf_occ_at(g, mip, G, C, ox, oy, oz, dx, dy, dz, tj));
            const unsigned b = coop_ballot(stop, lane);
            if (!done) {
                if (b) {
                    const int first = __ffs((int)b) - 1;
                    tmax = __shfl(tj, gbase + first, 64);
                    done = true;
                } else {
                    tcur = __shfl(tj, gbase + COOP - 1, 64) - mstep;
                }
            }
        }
    }
    if (!live || j != 0) return;
    bool mask = tmin < tmax;
    if (light_mask) mask = mask && !light_mask[idx];
    out_dirs[r * 3 + 0] = dx; out_dirs[r * 3 + 1] = dy; out_dirs[r * 3 + 2] = dz;
    out_minmax[r * 2 + 0] = tmin; out_minmax[r * 2 + 1] = tmax;
    out_mask[r] = mask ? 1 : 0;
    out_count[r] = mask ? (int32_t)((tmax - tmin) / step) : 0;
}

extern "C" int hrf_sampler_rays(const float* inverse_krs, const float* camera_origins, const uint8_t* landscape_modes,
                                const int64_t* ray_indices, const int64_t* grid_textures, const float* aabb,
                                const uint8_t* light_mask, int64_t num_rays, int grid_resolution, int image_width,
                                int image_height, float step, int use_occupancy,
                                float* out_dirs, float* out_minmax, uint8_t* out_mask, int32_t* out_count,
                                int32_t* workspace, hrf_stream_t stream)
{
    HRF_CHECK_ARG(num_rays >= 0, "negative num_rays");
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(inverse_krs && camera_origins && landscape_modes && ray_indices && aabb, "NULL input");
    HRF_CHECK_ARG(out_dirs && out_minmax && out_mask && out_count, "NULL output");
    HRF_CHECK_ARG(!use_occupancy || (grid_textures && grid_resolution > 0), "occupancy mode needs grids");
    HRF_CHECK_ARG(image_width > 0 && image_height > 0 && step > 0.0f, "bad image size / step");
    dim3 grid(hrf_blocks(num_rays, 256)), block(256);
    if (use_occupancy && workspace && grid_resolution % HRF_MIP2 == 0 && num_rays < ((int64_t)1 << 31)) {
        // two launches: the conservative block-mip test for every ray, then the exact march for the rays that passed it, packed
        // into full wavefronts (workspace: 1 counter + num_rays ray ids)
        if (hipMemsetAsync(workspace, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
            hrf_set_error("%s: hipMemsetAsync failed", __func__);
            return 2;
        }
        const unsigned pre_blocks = (unsigned)std::min<int64_t>(hrf_blocks(num_rays, 16),
                                                                 std::max<int64_t>(1024, hrf_blocks(num_rays, PRE_CAP - 16)));
        hipLaunchKernelGGL(k_sampler_prepass, dim3(pre_blocks), block, 0, (hipStream_t)stream, inverse_krs,
                           camera_origins, landscape_modes, ray_indices, grid_textures, aabb, num_rays, grid_resolution, image_width,
                           image_height, out_dirs, out_minmax, out_mask, out_count, workspace + 1, workspace);
        hipLaunchKernelGGL(k_sampler_rays_coop, dim3(hrf_blocks(num_rays * COOP, 256)), block, 0, (hipStream_t)stream,
                           inverse_krs, camera_origins, landscape_modes, ray_indices, grid_textures, aabb, light_mask,
                           num_rays, grid_resolution, image_width, image_height, step, out_dirs, out_minmax, out_mask,
                           out_count, (const int32_t*)(workspace + 1), (const int32_t*)workspace);
    } else if (use_occupancy)
        hipLaunchKernelGGL(k_sampler_rays_coop, dim3(hrf_blocks(num_rays * COOP, 256)), block, 0, (hipStream_t)stream,
                           inverse_krs, camera_origins, landscape_modes, ray_indices, grid_textures, aabb, light_mask,
                           num_rays, grid_resolution, image_width, image_height, step, out_dirs, out_minmax, out_mask,
                           out_count, (const int32_t*)nullptr, (const int32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_sampler_rays, grid, block, 0, (hipStream_t)stream, inverse_krs, camera_origins,
                           landscape_modes, ray_indices, grid_textures, aabb, light_mask, num_rays, grid_resolution,
                           image_width, image_height, step, out_dirs, out_minmax, out_mask, out_count);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan (single workgroup, serial carry over 4096-element chunks; n is O(rays))
// ------------------------------------------------------------------------------------------------
template <bool kU8>
__global__ __launch_bounds__(1024) void k_scan_exclusive(const void* __restrict__ in, int64_t n, int32_t* __restrict__ out)
{
    __shared__ int32_t wave_sums[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 4096) {
        int32_t v[4];
        int32_t local = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t i = base + (int64_t)tid * 4 + k;
            int32_t x = 0;
            if (i < n) x = kU8 ? (int32_t)((const uint8_t*)in)[i] : ((const int32_t*)in)[i];
            v[k] = local;
            local += x;
        }
        // wave inclusive scan of `local`
        int32_t incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_sums[wave] = incl;
        __syncthreads();
        int32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += wave_sums[w];
        int32_t total = 0;
        for (int w = 0; w < 16; ++w) total += wave_sums[w];
        const int32_t carry = carry_s;
        const int32_t excl = carry + wave_off + incl - local;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t i = base + (int64_t)tid * 4 + k;
            if (i < n) out[i] = excl + v[k];
        }
        __syncthreads();
        if (tid == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry_s;
}

// Multi-workgroup variant for long inputs: per-chunk local scans, a scan of the chunk totals, then the offsets.
// (Round 6 built the one-launch alternative -- a chained scan with decoupled look-back, 256-thread workgroups, ticket-ordered chunks,
// epoch-tagged state words -- to halve the 8.6 scan launches of a training step. It was correct (the scan tests ran on it) and the
// STEP got 2 % slower with it, alternated against this form on one trajectory (profiles/r06_ab_scan_priority_vectors.txt): the scans run
// on the sampler's side stream under the prune march, and workgroups that spin on their predecessors' state hold wavefront slots the
// march wants, where these two short launches just queue. Removed.)
template <bool kU8>
__global__ __launch_bounds__(1024) void k_scan_chunks(const void* __restrict__ in, int64_t n, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ chunk_sums)
{
    __shared__ int32_t wave_sums[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * 4096;
    int32_t v[4];
    int32_t local = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)tid * 4 + k;
        int32_t x = 0;
        if (i < n) x = kU8 ? (int32_t)((const uint8_t*)in)[i] : ((const int32_t*)in)[i];
        v[k] = local;
        local += x;
    }
    int32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int32_t wave_off = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wave_off += wave_sums[w];
        total += wave_sums[w];
    }
    const int32_t excl = wave_off + incl - local;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)tid * 4 + k;
        if (i < n) out[i] = excl + v[k];
    }
    if (tid == 0) chunk_sums[blockIdx.x] = total;
}

// Second (last) pass of the multi-workgroup scan: workgroup b sums the totals of the chunks before it (at most a few
// thousand values), adds that offset to its chunk and, for the last chunk, writes the grand total to out[n]. Replaces a
// single-workgroup scan of the totals + an add pass + a copy: two launches per scan instead of four stream operations
// (they sit on the critical path between the prune march and the size read-back).
__global__ __launch_bounds__(1024) void k_scan_finish(int64_t n, int32_t* __restrict__ out, const int32_t* __restrict__ chunk_sums,
                                                      int64_t chunks)
{
    __shared__ int32_t wave_sums[16];
    __shared__ int32_t s_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    int32_t part = 0;
    for (int64_t jj = tid; jj < b; jj += 1024) part += chunk_sums[jj];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) wave_sums[wave] = part;
    __syncthreads();
    if (tid == 0) {
        int32_t t = 0;
        for (int w = 0; w < 16; ++w) t += wave_sums[w];
        s_off = t;
        if (b == chunks - 1) out[n] = t + chunk_sums[b];
    }
    __syncthreads();
    const int32_t off = s_off;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = b * 4096 + (int64_t)k * 1024 + tid;
        if (i < n) out[i] += off;
    }
}


extern "C" int hrf_scan_exclusive(const void* in, int in_is_u8, int64_t n, int32_t* out, int32_t* workspace,
                                  hrf_stream_t stream)
{
    HRF_CHECK_ARG(n >= 0 && out != nullptr && (n == 0 || in != nullptr), "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int64_t chunks = (n + 4095) / 4096;
    if (workspace == nullptr || chunks <= 2) {
        if (in_is_u8) hipLaunchKernelGGL(k_scan_exclusive<true>, dim3(1), dim3(1024), 0, st, in, n, out);
        else hipLaunchKernelGGL(k_scan_exclusive<false>, dim3(1), dim3(1024), 0, st, in, n, out);
        HRF_CHECK_LAUNCH();
        return 0;
    }
    // workspace: chunk totals (callers size it 2 * chunks + 1 ints or more; only the first `chunks` are used)
    int32_t* sums = workspace;
    if (in_is_u8) hipLaunchKernelGGL(k_scan_chunks<true>, dim3((unsigned)chunks), dim3(1024), 0, st, in, n, out, sums);
    else hipLaunchKernelGGL(k_scan_chunks<false>, dim3((unsigned)chunks), dim3(1024), 0, st, in, n, out, sums);
    hipLaunchKernelGGL(k_scan_finish, dim3((unsigned)chunks), dim3(1024), 0, st, n, out, sums, chunks);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Ray compaction + gathers (ray_sampler.cu:258-266)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_rays(
    const int64_t* __restrict__ ray_indices, const uint8_t* __restrict__ mask, const int32_t* __restrict__ slot,
    const float* __restrict__ dirs_all, const float* __restrict__ minmax_all, const int32_t* __restrict__ count_all,
    const uint8_t* __restrict__ rgba_pool, const float* __restrict__ camera_origins,
    const int32_t* __restrict__ frame_numbers, const int32_t* __restrict__ camera_numbers, int64_t n,
    int64_t pixels_per_image, float* __restrict__ o_org, float* __restrict__ o_dir, float* __restrict__ o_rgba,
    int32_t* __restrict__ o_frame, int32_t* __restrict__ o_cam, float* __restrict__ o_mm, int32_t* __restrict__ o_cnt,
    int64_t* __restrict__ o_idx, const int32_t* __restrict__ cand_offset_all, int32_t* __restrict__ o_cand_offset)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !mask[r]) return;
    const int32_t s = slot[r];
    // masked-out rays have count 0, so the exclusive scan of the drawn rays' candidate counts, read at the surviving
    // rays, IS the exclusive scan over the compacted rays
    if (o_cand_offset) o_cand_offset[s] = cand_offset_all[r];
    const int64_t idx = ray_indices[r];
    const int64_t image = idx / pixels_per_image;  // torch::floor_divide(ray_indices, w*h)
    o_idx[s] = idx;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o_org[s * 3 + k] = camera_origins[image * 3 + k];
        o_dir[s * 3 + k] = dirs_all[r * 3 + k];
    }
    o_mm[s * 2 + 0] = minmax_all[r * 2 + 0];
    o_mm[s * 2 + 1] = minmax_all[r * 2 + 1];
    o_cnt[s] = count_all[r];
    if (o_frame) o_frame[s] = frame_numbers[image];
    if (o_cam) o_cam[s] = camera_numbers[image];
    if (o_rgba) {
        const uchar4 c = ((const uchar4*)rgba_pool)[idx];
        // (rgba.index(...) / 255.0f): fp32 true division (ray_sampler.cu:262)
        o_rgba[s * 4 + 0] = (float)c.x / 255.0f;
        o_rgba[s * 4 + 1] = (float)c.y / 255.0f;
        o_rgba[s * 4 + 2] = (float)c.z / 255.0f;
        o_rgba[s * 4 + 3] = (float)c.w / 255.0f;
    }
}

extern "C" int hrf_sampler_compact_rays(const int64_t* ray_indices, const uint8_t* mask, const int32_t* slot,
                                        const float* dirs_all, const float* minmax_all, const int32_t* count_all,
                                        const uint8_t* rgba_pool, const float* camera_origins,
                                        const int32_t* frame_numbers, const int32_t* camera_numbers,
                                        int64_t num_rays_in, int64_t pixels_per_image,
                                        float* out_origins, float* out_dirs, float* out_rgba, int32_t* out_frames,
                                        int32_t* out_cameras, float* out_minmax, int32_t* out_count,
                                        int64_t* out_ray_indices, const int32_t* cand_offset_all,
                                        int32_t* out_cand_offset, hrf_stream_t stream)
{
    if (num_rays_in == 0) return 0;
    HRF_CHECK_ARG(ray_indices && mask && slot && dirs_all && minmax_all && count_all && camera_origins, "NULL input");
    HRF_CHECK_ARG(out_origins && out_dirs && out_minmax && out_count && out_ray_indices, "NULL output");
    HRF_CHECK_ARG(!out_rgba || rgba_pool, "rgba requested without a pool");
    HRF_CHECK_ARG((!out_frames || frame_numbers) && (!out_cameras || camera_numbers), "frame/camera tables missing");
    HRF_CHECK_ARG(!out_cand_offset || cand_offset_all, "candidate offsets requested without the scan of count_all");
    hipLaunchKernelGGL(k_compact_rays, dim3(hrf_blocks(num_rays_in, 256)), dim3(256), 0, (hipStream_t)stream,
                       ray_indices, mask, slot, dirs_all, minmax_all, count_all, rgba_pool, camera_origins,
                       frame_numbers, camera_numbers, num_rays_in, pixels_per_image, out_origins, out_dirs, out_rgba,
                       out_frames, out_cameras, out_minmax, out_count, out_ray_indices, cand_offset_all, out_cand_offset);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Stage 2: per-sample distance + occupancy filter + compaction (ray_sampler.cu:149-194, 322-323)
// One wavefront per ray; lanes take 64 consecutive candidate samples; __ballot + prefix popcount give
// each survivor its slot, so the output is written sorted by (ray, distance) without any global scan
// over the (much larger) candidate set.
// ------------------------------------------------------------------------------------------------
template <bool kOcc, bool kWrite>
__global__ __launch_bounds__(256) void k_sampler_samples(
    const int64_t* __restrict__ ray_indices, const int64_t* __restrict__ grid_textures,
    const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ minmax,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offsets, int64_t num_rays,
    const int32_t* __restrict__ num_rays_dev, int64_t pixels_per_image, int G, float step,
    int32_t* __restrict__ out_kept, float* __restrict__ out_t, int32_t* __restrict__ out_ray, int64_t capacity)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= num_rays) return;
    if (num_rays_dev && r >= *num_rays_dev) {  // slot beyond the device-side ray count (host passed an upper bound)
        if (out_kept && lane == 0) out_kept[r] = 0;
        return;
    }
    const int32_t cnt = count[r];
    const float tmin = minmax[r * 2];
    const float ox = origins[r * 3 + 0], oy = origins[r * 3 + 1], oz = origins[r * 3 + 2];
    const float dx = dirs[r * 3 + 0], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
    hrf_gbytes g = nullptr;
    if (kOcc && cnt > 0) g = (hrf_gbytes)(uintptr_t)grid_textures[ray_indices[r] / pixels_per_image];
    const int C = (G % HRF_MIP == 0) ? G / HRF_MIP : 0;
    hrf_gbytes mip = (kOcc && C && g) ? g + (size_t)G * G * G : nullptr;
    int32_t base = kWrite ? offsets[r] : 0;
    int32_t kept = 0;
    for (int32_t c0 = 0; c0 < cnt; c0 += 64) {
        const int32_t local = c0 + lane;
        const float t = tmin + (float)local * step;
        bool keep = local < cnt;
        if (kOcc && keep) keep = hrf_occ_at(g, mip, G, C, ox, oy, oz, dx, dy, dz, t);
        const unsigned long long b = __ballot(keep);
        if (kWrite) {
            const int pre = __popcll(b & ((1ull << lane) - 1ull));
            if (keep && (int64_t)base + kept + pre < capacity) {  // capacity: size of out_t / out_ray
                out_t[base + kept + pre] = t;
                if (out_ray) out_ray[base + kept + pre] = (int32_t)r;
            }
        }
        kept += __popcll(b);
    }
    // Single-pass form (offsets = exclusive scan of the CANDIDATE counts, out_kept given together with out_t): every ray
    // owns the slot range of its candidates and fills a prefix of it, so the occupancy predicate is evaluated once and
    // no scan over the surviving counts is needed; the consumer reads [offsets[r], offsets[r] + out_kept[r]).
    if (out_kept && lane == 0) out_kept[r] = kept;
}

extern "C" int hrf_sampler_samples(const int64_t* ray_indices, const int64_t* grid_textures, const float* origins,
                                   const float* dirs, const float* minmax, const int32_t* count,
                                   const int32_t* offsets, int64_t num_rays, const int32_t* num_rays_dev,
                                   int64_t pixels_per_image, int grid_resolution, float step, int use_occupancy,
                                   int32_t* out_kept, float* out_t, int32_t* out_ray, int64_t capacity,
                                   hrf_stream_t stream)
{
    if (num_rays == 0) return 0;
    HRF_CHECK_ARG(origins && dirs && minmax && count, "NULL input");
    HRF_CHECK_ARG(!use_occupancy || (ray_indices && grid_textures), "occupancy mode needs ray_indices and grids");
    const bool write = out_t != nullptr;
    HRF_CHECK_ARG(write ? (offsets != nullptr) : (out_kept != nullptr), "inconsistent pass arguments");
    dim3 grid(hrf_blocks(num_rays * 64, 256)), block(256);
#define HRF_LAUNCH_SS(OCC, WR)                                                                                      \
    hipLaunchKernelGGL((k_sampler_samples<OCC, WR>), grid, block, 0, (hipStream_t)stream, ray_indices, grid_textures, \
                       origins, dirs, minmax, count, offsets, num_rays, num_rays_dev, pixels_per_image, grid_resolution, step, \
                       out_kept, out_t, out_ray, capacity)
    if (use_occupancy) { if (write) HRF_LAUNCH_SS(true, true); else HRF_LAUNCH_SS(true, false); }
    else { if (write) HRF_LAUNCH_SS(false, true); else HRF_LAUNCH_SS(false, false); }
#undef HRF_LAUNCH_SS
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Pool replacement (actorshq/dataset/data_loader.py:396-511, _replace_next_buffer_entry / _load_and_copy_camera_frame_data)
// for a capture that is resident in HBM: refilling k pool slots is ONE launch -- k image copies (2.3 MB each at 4x)
// plus the per-slot camera tables the sampler reads -- instead of a JPEG decode, a CPU->GPU copy and six small tensor
// writes per slot under the loader's lock. spec[e] = { slot, camera index in the capture, frame index in the capture,
// camera number, frame number }.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_replace(
    const int32_t* __restrict__ spec, const uint8_t* __restrict__ capture, int64_t pixels, int capture_frames,
    uint8_t* __restrict__ pool, const float* __restrict__ all_inverse_krs, const float* __restrict__ all_origins,
    const uint8_t* __restrict__ all_landscape, const int64_t* __restrict__ grid_by_frame,
    int32_t* __restrict__ frame_numbers, int32_t* __restrict__ camera_numbers, uint8_t* __restrict__ landscape,
    float* __restrict__ inverse_krs, float* __restrict__ origins, int64_t* __restrict__ grid_textures)
{
    const int e = blockIdx.y;
    const int slot = spec[e * 5 + 0], ci = spec[e * 5 + 1], fi = spec[e * 5 + 2], cam = spec[e * 5 + 3], frame = spec[e * 5 + 4];
    const uint8_t* src8 = capture + ((size_t)ci * capture_frames + fi) * (size_t)pixels * 4;
    uint8_t* dst8 = pool + (size_t)slot * (size_t)pixels * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((pixels & 3) == 0) {   // every image starts on a 16-byte boundary (bases come from the allocator)
        const uint4* src = (const uint4*)src8;
        uint4* dst = (uint4*)dst8;
        for (int64_t i = tid; i < (pixels >> 2); i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = tid; i < pixels; i += stride) ((uint32_t*)dst8)[i] = ((const uint32_t*)src8)[i];
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < 9) inverse_krs[slot * 9 + threadIdx.x] = all_inverse_krs[cam * 9 + threadIdx.x];
        if (threadIdx.x < 3) origins[slot * 3 + threadIdx.x] = all_origins[cam * 3 + threadIdx.x];
        if (threadIdx.x == 0) {
            frame_numbers[slot] = frame;
            camera_numbers[slot] = cam;
            landscape[slot] = all_landscape[cam];
            if (grid_textures) grid_textures[slot] = grid_by_frame[fi];
        }
    }
}

extern "C" int hrf_pool_replace(const int32_t* spec, int count, const uint8_t* capture, int64_t pixels_per_image,
                                int capture_frames, uint8_t* pool, const float* all_inverse_krs, const float* all_origins,
                                const uint8_t* all_landscape, const int64_t* grid_by_frame, int32_t* frame_numbers,
                                int32_t* camera_numbers, uint8_t* landscape_modes, float* inverse_krs,
                                float* camera_origins, int64_t* grid_textures, hrf_stream_t stream)
{
    if (count == 0) return 0;
    HRF_CHECK_ARG(spec && capture && pool && all_inverse_krs && all_origins && all_landscape, "NULL input");
    HRF_CHECK_ARG(frame_numbers && camera_numbers && landscape_modes && inverse_krs && camera_origins, "NULL table");
    HRF_CHECK_ARG(!grid_textures || grid_by_frame, "grid handles requested without the per-frame table");
    HRF_CHECK_ARG(count > 0 && count <= 65535 && pixels_per_image > 0 && capture_frames > 0, "bad sizes");
    HRF_CHECK_ARG((((uintptr_t)capture | (uintptr_t)pool) & 15u) == 0, "capture / pool must be 16-byte aligned");
    unsigned bx = hrf_blocks(pixels_per_image >> 2, 256 * 8);
    if (bx == 0) bx = 1;
    hipLaunchKernelGGL(k_pool_replace, dim3(bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, spec, capture,
                       pixels_per_image, capture_frames, pool, all_inverse_krs, all_origins, all_landscape, grid_by_frame,
                       frame_numbers, camera_numbers, landscape_modes, inverse_krs, camera_origins, grid_textures);
    HRF_CHECK_LAUNCH();
    return 0;
}
