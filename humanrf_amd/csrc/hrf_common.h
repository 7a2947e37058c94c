// Shared device helpers for libhrf_hip.so (gfx950 only). Built with -ffp-contract=off so that every
// fp32 expression below rounds exactly as written (the sampler must be bit-identical to oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/hrf.h"

#define HRF_WAVE 64

void hrf_set_error(const char* fmt, ...);

#define HRF_CHECK_ARG(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            hrf_set_error("%s: %s", __func__, msg);                \
            return 1;                                              \
        }                                                          \
    } while (0)

#define HRF_CHECK_LAUNCH()                                                           \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            hrf_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return 2;                                                                \
        }                                                                            \
    } while (0)

static inline unsigned hrf_blocks(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------------------------
// Occupancy volume: software restatement of the reference's texture fetch predicate
// (occupancy_grid.cu:28-35). Returns tex3D<float>(grid, x, y, z) > 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void hrf_tex_axis(float c, int G, int& lo, int& hi, bool& a0, bool& a1)
{
    float xb = c * (float)G - 0.5f;
    float fl = floorf(xb);
    float fr = xb - fl;
    int aq = (int)floorf(fr * 256.0f + 0.5f);
    float flc = fl < -1.0f ? -1.0f : (fl > (float)G ? (float)G : fl);
    if (!(fl == fl)) { flc = -1.0f; aq = 0; }
    int i = (int)flc;
    lo = i < 0 ? 0 : (i > G - 1 ? G - 1 : i);
    hi = (i + 1) < 0 ? 0 : ((i + 1) > G - 1 ? G - 1 : (i + 1));
    a0 = aq < 256;
    a1 = aq > 0;
}

// Occupancy volumes are reached through 64-bit handles stored in memory (the reference's texture objects); a pointer that
// comes out of memory is generic, and the compiler would emit flat_load for every texel fetch of the march. The handle is
// a device allocation, so it is typed as a global-address-space pointer: global_load_ubyte.
typedef const __attribute__((address_space(1))) uint8_t* hrf_gbytes;

__device__ __forceinline__ bool hrf_tex_gt0(hrf_gbytes g, int G, float x, float y, float z)
{
    int x0, x1, y0, y1, z0, z1;
    bool ax0, ax1, ay0, ay1, az0, az1;
    hrf_tex_axis(x, G, x0, x1, ax0, ax1);
    hrf_tex_axis(y, G, y0, y1, ay0, ay1);
    hrf_tex_axis(z, G, z0, z1, az0, az1);
    const size_t GG = (size_t)G * G;
    const size_t r00 = (size_t)z0 * GG + (size_t)y0 * G, r01 = (size_t)z0 * GG + (size_t)y1 * G;
    const size_t r10 = (size_t)z1 * GG + (size_t)y0 * G, r11 = (size_t)z1 * GG + (size_t)y1 * G;
    // Issue all eight byte loads unconditionally (independent -> one latency), mask afterwards.
    unsigned v = 0;
    v |= (ax0 && ay0 && az0) ? g[r00 + x0] : 0;
    v |= (ax1 && ay0 && az0) ? g[r00 + x1] : 0;
    v |= (ax0 && ay1 && az0) ? g[r01 + x0] : 0;
    v |= (ax1 && ay1 && az0) ? g[r01 + x1] : 0;
    v |= (ax0 && ay0 && az1) ? g[r10 + x0] : 0;
    v |= (ax1 && ay0 && az1) ? g[r10 + x1] : 0;
    v |= (ax0 && ay1 && az1) ? g[r11 + x0] : 0;
    v |= (ax1 && ay1 && az1) ? g[r11 + x1] : 0;
    return v != 0;
}

// Coarse occupancy mip (library-built, stored right after each ring volume): one byte per 4^3 block, set when
// any texel of the block EXPANDED BY ONE TEXEL on every side is non-zero. A fetch at texture coordinate c
// touches texels floor(c*G - 0.5) + {0,1} (clamped), all inside the expanded block floor(c*G/4), so
// "mip byte == 0" implies "tex3D(...) > 0 is false" exactly -- the march skips the eight texel loads there
// and still takes the same fp32 steps, keeping tmin / tmax bit-identical to the plain march.
#define HRF_MIP 4
__device__ __forceinline__ int hrf_mip_axis(float c, int G, int C)
{
    float f = (c * (float)G) * 0.25f;
    f = fminf(fmaxf(f, 0.0f), (float)(C - 1));  // NaN -> 0
    return (int)f;
}

__device__ __forceinline__ bool hrf_occ_at(hrf_gbytes g, hrf_gbytes mip, int G, int C,
                                           float ox, float oy, float oz, float dx, float dy, float dz, float t)
{
    float px = (ox + dx * t) + 0.5f;
    float py = (oy + dy * t) + 0.5f;
    float pz = (oz + dz * t) + 0.5f;
    if (mip) {
        const int cx = hrf_mip_axis(px, G, C), cy = hrf_mip_axis(py, G, C), cz = hrf_mip_axis(pz, G, C);
        if (mip[((size_t)cz * C + cy) * C + cx] == 0) return false;
    }
    return hrf_tex_gt0(g, G, px, py, pz);
}

// ---------------------------------------------------------------------------------------------
// tcnn HashGrid indexing
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hrf_grid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res,
                                                   uint32_t size, bool hashed)
{
    uint32_t idx = hashed ? (x ^ (y * 2654435761u) ^ (z * 805459861u)) : (x + y * res + z * res * res);
    return idx % size;
}

// 1-D vector grid tap (tensor_composition.cu:37-45): coord*Rv - 0.5, clamped taps.
__device__ __forceinline__ void hrf_vec_tap(float c, int Rv, int& c0, int& c1, float& fr)
{
    float coord = c * (float)Rv - 0.5f;
    float fl = floorf(coord);
    fr = coord - fl;
    // both taps clamped to [0, Rv-1] (the reference clamps one side each and reads OOB outside [0,1])
    c0 = (int)fminf(fmaxf(fl, 0.0f), (float)(Rv - 1));
    c1 = (int)fminf(fmaxf(fl + 1.0f, 0.0f), (float)(Rv - 1));
}

__device__ __forceinline__ float hrf_h2f(__half h) { return __half2float(h); }

// The reference's modules hand each other HALF gradient tensors at the GradScaler's scale (decomposition4d.py:8-39: the compose
// op's d_out and its four per-encoding outputs; tcnn modules: dL/d(output) arrives as the dtype of the output), while the
// fused backward carries fp32 values at `b` times that scale (b = tcnn's internal loss_scale, 128). x -> the value a half
// tensor at the reference's scale would hold, brought back to the fused scale; b = 0: x unchanged. Contributions below
// 2^-25 / scale round to zero, as they do in the reference.
// x / b as the half the reference's gradient tensor holds: the fp32 product x * (1 / b), THEN one rounding to half. (The empty asm
// pins the fp32 product: left alone, the compiler folds multiply and conversion into v_fma_mixlo_f16 in some kernels and not in
// others, and that instruction rounds the exact product to half once -- a different half in rare cases, so two kernels that
// must agree on the boundary would not.)
__device__ __forceinline__ __half hrf_boundary_half(float x, float inv_b)
{
    float y = x * inv_b;
    asm("" : "+v"(y));          // (not volatile: the value is pinned, the statement may still be scheduled freely)
    return __float2half_rn(y);
}

__device__ __forceinline__ float hrf_through_half(float x, float b, float inv_b)
{
    return b > 0.0f ? __half2float(hrf_boundary_half(x, inv_b)) * b : x;
}

// ---------------------------------------------------------------------------------------------
// Counter-based uniform numbers in [0,1) with 24 random bits (the format torch.rand produces): value number `idx` of
// the stream `seed`. Used where the reference calls torch.rand_like inside the step (the jitter of prune_samples,
// volume_rendering.py:63-64): the kernel that consumes the number computes it, nothing is stored or launched.
// hrf_uniform_fill writes the same values to memory so that tests can feed them to the oracle.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hrf_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float hrf_uniform01(uint32_t seed, uint32_t idx)
{
    const uint32_t h = hrf_mix32(idx ^ hrf_mix32(seed ^ 0x9e3779b9u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
