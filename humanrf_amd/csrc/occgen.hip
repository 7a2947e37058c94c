// Occupancy grids from foreground masks (visual-hull carving) for gfx950: the step BEFORE the hot path that produces
// the grids it consumes. Replaces generate_from_masks_kernel
// (actorshq/toolbox/native/occupancy_grid_generation.cu:16-80) and the cv2.dilate of the driver
// (actorshq/toolbox/generate_occupancy_grids_from_masks.py:64-77).
//
// One thread per voxel, x fastest, as in the reference: neighbouring lanes project to neighbouring pixels, so the
// four mask bytes per camera are mostly shared cache lines; the projection matrices are wavefront-uniform (scalar
// loads -- the reference keeps them in __constant__ memory, which made the function non-re-entrant). The camera loop
// ends for the whole wavefront as soon as every lane has decided.
//
// Arithmetic fixed here and in the CPU restatement used by the tests (the reference builds with --use_fast_math): IEEE fp32, no FMA
// contraction (this library is built with -ffp-contract=off), GLM's operand order for mat4 * vec4
// ((m0*x + m1*y) + (m2*z + m3*w)), correctly rounded division, float -> int conversion truncating toward zero,
// saturating, NaN -> 0 (what both cvt.rzi.s32.f32 and v_cvt_i32_f32 do).
#include "hrf_common.h"

__device__ __forceinline__ int hrf_f2i_rz(float v) { return __float2int_rz(v); }  // v_cvt_i32_f32: trunc, saturating, NaN -> 0

__global__ __launch_bounds__(256) void k_grid_from_masks(const uint8_t* __restrict__ masks, const float* __restrict__ proj,
                                                         const uint8_t* __restrict__ landscape, int threshold,
                                                         int num_cameras, int G, int width, int height,
                                                         uint8_t* __restrict__ grid)
{
    const int64_t voxel = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)G * G * G;
    const bool live = voxel < total;
    const int gx = (int)(voxel % G), gy = (int)((voxel / G) % G), gz = (int)((voxel / G) / G);
    const float den = (float)(G - 1);
    const float vx = (float)gx / den - 0.5f, vy = (float)gy / den - 0.5f, vz = (float)gz / den - 0.5f;  // :37
    const size_t P = (size_t)width * height;
    int covered = 0;
    bool in_hull = false, done = !live;
    for (int c = 0; c < num_cameras; ++c) {
        if (__all(done)) break;
        if (done) continue;
        const float* m = proj + (size_t)c * 16;  // column-major 4x4
        const bool land = landscape[c] != 0;
        const int cw = land ? width : height, ch = land ? height : width;
        const float px = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * 1.0f);
        const float py = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * 1.0f);
        const float pz = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * 1.0f);
        const int x = hrf_f2i_rz(px / pz), y = hrf_f2i_rz(py / pz);
        if (x >= 0 && x < cw && y >= 0 && y < ch) {
            const int x1 = min(x + 1, cw - 1), y1 = min(y + 1, ch - 1);
            const uint8_t* mk = masks + (size_t)c * P;
            const bool empty = mk[(size_t)x + (size_t)y * cw] == 0 && mk[(size_t)x1 + (size_t)y * cw] == 0 &&
                               mk[(size_t)x + (size_t)y1 * cw] == 0 && mk[(size_t)x1 + (size_t)y1 * cw] == 0;
            if (empty) {
                const int rest = num_cameras - c - 1;
                if (covered + rest < threshold) done = true;  // cannot reach the threshold any more (:64-68)
            } else {
                ++covered;
                in_hull = covered >= threshold;
                if (in_hull) done = true;
            }
        }
    }
    if (live) grid[voxel] = in_hull ? 255 : 0;  // [z][y][x]
}

extern "C" int hrf_occgrid_from_masks(const uint8_t* masks, const float* projection_matrices,
                                      const uint8_t* landscape_modes, int camera_coverage_threshold, int num_cameras,
                                      int grid_resolution, int width, int height, uint8_t* out_grid,
                                      hrf_stream_t stream)
{
    HRF_CHECK_ARG(masks && projection_matrices && landscape_modes && out_grid, "NULL argument");
    HRF_CHECK_ARG(num_cameras > 0 && grid_resolution > 1 && width > 0 && height > 0, "bad sizes");
    const int64_t total = (int64_t)grid_resolution * grid_resolution * grid_resolution;
    hipLaunchKernelGGL(k_grid_from_masks, dim3(hrf_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, masks,
                       projection_matrices, landscape_modes, camera_coverage_threshold, num_cameras, grid_resolution,
                       width, height, out_grid);
    HRF_CHECK_LAUNCH();
    return 0;
}

// cv2.dilate(mask, ones((k, k)), iterations=1): maximum over a k x k window anchored at (k/2, k/2), pixels outside the
// image ignored. One thread per output pixel; rows of the window are contiguous bytes.
__global__ __launch_bounds__(256) void k_mask_dilate(const uint8_t* __restrict__ in, int width, int height, int k, int64_t images,
                                                     uint8_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t P = (int64_t)width * height;
    if (i >= P * images) return;
    const int64_t img = i / P;
    const int x = (int)(i % width), y = (int)((i % P) / width);
    const int a = k / 2;
    const uint8_t* src = in + img * P;
    uint8_t v = 0;
    for (int dy = -a; dy < k - a; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= height) continue;
        for (int dx = -a; dx < k - a; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= width) continue;
            const uint8_t s = src[(size_t)yy * width + xx];
            v = s > v ? s : v;
        }
    }
    out[i] = v;
}

extern "C" int hrf_mask_dilate(const uint8_t* masks, int width, int height, int kernel_size, int64_t num_images,
                               uint8_t* out, hrf_stream_t stream)
{
    HRF_CHECK_ARG(masks && out && masks != out, "NULL or aliased argument");
    HRF_CHECK_ARG(width > 0 && height > 0 && kernel_size > 0 && num_images >= 0, "bad sizes");
    if (num_images == 0) return 0;
    hipLaunchKernelGGL(k_mask_dilate, dim3(hrf_blocks((int64_t)width * height * num_images, 256)), dim3(256), 0,
                       (hipStream_t)stream, masks, width, height, kernel_size, num_images, out);
    HRF_CHECK_LAUNCH();
    return 0;
}
