// oracle/ref_cuda/driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the reference's own CUDA kernels (source text cut out of /root/reference at build time, see shim.h and
// oracle/build_ref_cuda.py) on the host, one "thread" after the other, behind a C ABI that tests/test_cpu_ref_cuda.py calls with
// ctypes -> oracle/_ref/libhrf_refcuda.so. The entry points mirror what the reference's host functions hand their kernels
// (ray_sampler.cu:196-325 get_data<>, tensor_composition.cu:120-240), with plain pointers instead of tensors.
#include "shim.h"
#include <vector>

#include "ray_sampler_device.inc"          // generated: ray_sampler.cu, kAabb .. compute_sample_distances_kernel
#include "tensor_composition_device.inc"   // generated: tensor_composition.cu, the two kernels
#include "occupancy_grid_generation_device.inc"   // generated: occupancy_grid_generation.cu, constants + the carving kernel

namespace {
constexpr unsigned kBlock = 256;
template <class F>
void for_each_thread(size_t n, F f)
{
    blockDim.x = kBlock;
    for (size_t i = 0; i < n; ++i) {
        blockIdx.x = (unsigned)(i / kBlock);
        threadIdx.x = (unsigned)(i % kBlock);
        f();
    }
}
struct Textures {
    std::vector<ShimTexture> tex;
    std::vector<int64_t> handles;
    Textures(const uint8_t* const* grids, int num_images, int G)
    {
        tex.resize((size_t)num_images);
        handles.resize((size_t)num_images, 0);
        for (int i = 0; i < num_images; ++i) {
            tex[i].texels = grids ? grids[i] : nullptr;
            tex[i].resolution = G;
            handles[i] = (int64_t)(uintptr_t)&tex[i];
        }
    }
};
}  // namespace

// compute_minmax_kernel<occupancy> over num_rays pixel ids (ray_sampler.cu:80-147, launched at :268-281)
extern "C" void ref_compute_minmax(int occupancy, const float* inverse_krs, const float* camera_origins, const uint8_t* landscape,
                                   const int64_t* ray_indices, const uint8_t* const* grids, const float* aabb, int64_t num_rays,
                                   int num_images, int grid_resolution, int width, int height, float* out_dirs, float* out_minmax,
                                   uint8_t* out_mask)
{
    kAabb[0] = glm::vec3(aabb[0], aabb[1], aabb[2]);     // cudaMemcpyToSymbol(kAabb, ...) of ray_sampler.cu:226
    kAabb[1] = glm::vec3(aabb[3], aabb[4], aabb[5]);
    Textures tx(grids, num_images, grid_resolution);
    static_assert(sizeof(bool) == 1, "bool accessors alias uint8 buffers");
    const size_t R = (size_t)num_rays, B = (size_t)num_images;
    auto a_ikr = shim_accessor<float, 3>(inverse_krs, {B, 3, 3});
    auto a_org = shim_accessor<float, 2>(camera_origins, {B, 3});
    auto a_land = shim_accessor<bool, 1>(reinterpret_cast<const bool*>(landscape), {B});
    auto a_idx = shim_accessor<int64_t, 1>(ray_indices, {R});
    auto a_tex = shim_accessor<int64_t, 1>(tx.handles.data(), {B});
    auto a_dirs = shim_accessor<float, 2>(out_dirs, {R, 3});
    auto a_mm = shim_accessor<float, 2>(out_minmax, {R, 2});
    auto a_mask = shim_accessor<bool, 1>(reinterpret_cast<bool*>(out_mask), {R});
    for_each_thread(R, [&] {
        if (occupancy) compute_minmax_kernel<true>(a_ikr, a_org, a_land, a_idx, a_tex, (int)num_rays, grid_resolution, width, height, a_dirs, a_mm, a_mask);
        else compute_minmax_kernel<false>(a_ikr, a_org, a_land, a_idx, a_tex, (int)num_rays, grid_resolution, width, height, a_dirs, a_mm, a_mask);
    });
}

// compute_sample_distances_kernel<occupancy> (ray_sampler.cu:149-194, launched at :299-316): ray_indices = the pixel ids of the
// COMPACTED rays, end_index_per_ray = inclusive cumsum of the per-ray counts, ray_of_sample = repeat_interleave of the ray numbers
extern "C" void ref_sample_distances(int occupancy, const int64_t* ray_indices, const uint8_t* const* grids, int num_images,
                                     int grid_resolution, const float* minmaxes, const float* origins, const float* dirs,
                                     const int32_t* end_index_per_ray, const int32_t* ray_of_sample, int64_t num_rays,
                                     int64_t num_samples, int num_pixels_per_camera, float step, float* out_t, uint8_t* out_keep)
{
    Textures tx(grids, num_images, grid_resolution);
    const size_t R = (size_t)num_rays, N = (size_t)num_samples, B = (size_t)num_images;
    auto a_idx = shim_accessor<int64_t, 1>(ray_indices, {R});
    auto a_tex = shim_accessor<int64_t, 1>(tx.handles.data(), {B});
    auto a_mm = shim_accessor<float, 2>(minmaxes, {R, 2});
    auto a_org = shim_accessor<float, 2>(origins, {R, 3});
    auto a_dir = shim_accessor<float, 2>(dirs, {R, 3});
    auto a_end = shim_accessor<int, 1>(end_index_per_ray, {R});
    auto a_ros = shim_accessor<int, 1>(ray_of_sample, {N});
    auto a_t = shim_accessor<float, 1>(out_t, {N});
    auto a_keep = shim_accessor<bool, 1>(reinterpret_cast<bool*>(out_keep), {N});
    for_each_thread(N, [&] {
        if (occupancy) compute_sample_distances_kernel<true>(a_idx, a_tex, a_mm, a_org, a_dir, a_end, a_ros, (int)num_samples, num_pixels_per_camera, step, a_t, a_keep);
        else compute_sample_distances_kernel<false>(a_idx, a_tex, a_mm, a_org, a_dir, a_end, a_ros, (int)num_samples, num_pixels_per_camera, step, a_t, a_keep);
    });
}

// compose_tensors_forward_kernel (tensor_composition.cu:9-57): halves as uint16 bit patterns
extern "C" void ref_compose_forward(const uint16_t* xyz, const uint16_t* xyt, const uint16_t* yzt, const uint16_t* xzt,
                                    const float* vectors, const float* xyzt, int64_t num_samples, int feature_dim, int resolution,
                                    uint16_t* out)
{
    const size_t N = (size_t)num_samples, F = (size_t)feature_dim, V = (size_t)resolution;
    auto h = [&](const uint16_t* p) { return shim_accessor<at::Half, 2>(reinterpret_cast<const at::Half*>(p), {N, F}); };
    auto a_vec = shim_accessor<float, 3>(vectors, {4, V, F});
    auto a_x = shim_accessor<float, 2>(xyzt, {N, 4});
    auto a_out = h(out);
    auto a0 = h(xyz), a1 = h(xyt), a2 = h(yzt), a3 = h(xzt);
    for_each_thread(N * F, [&] { compose_tensors_forward_kernel(a0, a1, a2, a3, a_vec, a_x, (int)num_samples, feature_dim, resolution, a_out); });
}

// compose_tensors_backward_kernel (tensor_composition.cu:59-118); d_vectors must come zeroed (torch::zeros_like, :190)
extern "C" void ref_compose_backward(const uint16_t* xyz, const uint16_t* xyt, const uint16_t* yzt, const uint16_t* xzt,
                                     const float* vectors, const float* xyzt, const uint16_t* d_out, int64_t num_samples,
                                     int feature_dim, int resolution, uint16_t* d_xyz, uint16_t* d_xyt, uint16_t* d_yzt,
                                     uint16_t* d_xzt, float* d_vectors)
{
    const size_t N = (size_t)num_samples, F = (size_t)feature_dim, V = (size_t)resolution;
    auto h = [&](const uint16_t* p) { return shim_accessor<at::Half, 2>(reinterpret_cast<const at::Half*>(p), {N, F}); };
    auto a_vec = shim_accessor<float, 3>(vectors, {4, V, F});
    auto a_dvec = shim_accessor<float, 3>(d_vectors, {4, V, F});
    auto a_x = shim_accessor<float, 2>(xyzt, {N, 4});
    auto a0 = h(xyz), a1 = h(xyt), a2 = h(yzt), a3 = h(xzt), a_do = h(d_out);
    auto g0 = h(d_xyz), g1 = h(d_xyt), g2 = h(d_yzt), g3 = h(d_xzt);
    for_each_thread(N * F, [&] {
        compose_tensors_backward_kernel(a0, a1, a2, a3, a_vec, a_x, a_do, (int)num_samples, feature_dim, resolution, g0, g1, g2, g3, a_dvec);
    });
}

// generate_from_masks_kernel (occupancy_grid_generation.cu:16-80, launched at :106-115): masks (C, W*H) uint8, proj (C,16) floats as the
// host function copies them into kProjectionMatrices (:100), landscape (C) bool
extern "C" int ref_grid_from_masks(const uint8_t* masks, const float* proj, const uint8_t* landscape, int threshold, int num_cameras,
                                   int grid_resolution, int width, int height, uint8_t* grid)
{
    if (num_cameras > kMaxNumCameras) return 1;
    std::memcpy(static_cast<void*>(kProjectionMatrices), proj, (size_t)num_cameras * sizeof(glm::mat4));      // cudaMemcpyToSymbol, :100-101
    for (int c = 0; c < num_cameras; ++c) kLandscapeModes[c] = landscape[c] != 0;
    const size_t G = (size_t)grid_resolution;
    auto a_masks = shim_accessor<uint8_t, 2>(masks, {(size_t)num_cameras, (size_t)width * (size_t)height});
    auto a_grid = shim_accessor<uint8_t, 3>(grid, {G, G, G});
    for_each_thread(G * G * G, [&] {
        generate_from_masks_kernel(a_masks, threshold, num_cameras, grid_resolution, width, height, a_grid);
    });
    return 0;
}
