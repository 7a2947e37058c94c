// oracle/ref_cuda/shim.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// What lets g++ compile the DEVICE code of the reference's own CUDA sources on the host, unmodified, where they lie under
// /root/reference (oracle/build_ref_cuda.py cuts the device functions and kernel bodies out of the .cu files at build time --
// the host functions around them use <<<...>>> launches and torch's C++ API -- into oracle/_ref/gen/, git-ignored):
//   actorshq/dataset/native/ray_sampler.cu:9-194                       kAabb, compute_aabb_minmax, compute_occupancy_minmax,
//                                                                      compute_minmax_kernel, compute_sample_distances_kernel
//   humanrf/scene_representation/native/tensor_composition.cu:9-118    compose_tensors_forward_kernel / _backward_kernel
//   actorshq/toolbox/native/occupancy_grid_generation.cu:10-80         kProjectionMatrices, kLandscapeModes, generate_from_masks_kernel
// Everything here is a stand-in for a header those files include and that is absent from this image: the CUDA runtime
// (qualifiers, blockIdx / threadIdx, tex3D, atomicAdd, __half conversions), torch's PackedTensorAccessor, at::Half, and the
// few GLM types and functions ray_sampler.cu uses. Nothing is copied from the reference or from those libraries; GLM's
// semantics are restated from its published definitions [UPSTREAM-KNOWLEDGE, GLM 0.9.9]:
//   operator/(float, vec3), vec3 - / * / + vec3, vec3 * float, vec3 + float: component-wise
//   min(x, y) = (y < x) ? y : x, max(x, y) = (x < y) ? y : x (component-wise for vectors)
//   dot(a, b) = (a.x b.x + a.y b.y) + a.z b.z; inversesqrt(x) = 1 / sqrt(x); normalize(v) = v * inversesqrt(dot(v, v))
//   mat3 is column-major; mat3 * vec3: row r = (m[0][r] v.x + m[1][r] v.y) + m[2][r] v.z
// The texture fetch is the ONE definition the whole build uses (oracle/sampler_oracle.c:orc_tex_gt0, linked in): tex3D returns
// 1.0f where that predicate holds, 0.0f elsewhere -- the sampler only ever tests "> 0". Built with -ffp-contract=off and no
// fast-math (the reference's --use_fast_math is not reproducible across vendors; DESIGN.md section 2 states the fixed arithmetic).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __constant__
#define __host__
#define __restrict__

struct shim_dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local shim_dim3 blockIdx, blockDim, threadIdx;

// ---------------------------------------------------------------- texture objects
typedef unsigned long long cudaTextureObject_t;
struct ShimTexture { const uint8_t* texels; int resolution; };   // (G,G,G) uint8 [z][y][x]
extern "C" int orc_tex_gt0(const uint8_t* g, int G, float x, float y, float z);   // oracle/sampler_oracle.c
template <class T>
static inline T tex3D(cudaTextureObject_t handle, float x, float y, float z)
{
    const ShimTexture* t = reinterpret_cast<const ShimTexture*>(static_cast<uintptr_t>(handle));
    return orc_tex_gt0(t->texels, t->resolution, x, y, z) ? (T)1 : (T)0;
}

// ---------------------------------------------------------------- half
namespace at {
struct Half { uint16_t bits; };
}
static inline float __half2float(at::Half h)
{
    const uint32_t s = (uint32_t)(h.bits & 0x8000u) << 16, e = (h.bits >> 10) & 31u, m = h.bits & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else {   // subnormal: value = m * 2^-24, exactly representable
            float f = (float)m * 5.9604644775390625e-08f;
            std::memcpy(&u, &f, 4);
            u |= s;
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline at::Half __float2half(float f)   // round to nearest even, like the hardware conversion
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    at::Half h;
    if (a >= 0x7f800000u) { h.bits = (uint16_t)(s | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0u)); return h; }
    if (a >= 0x477ff000u) { h.bits = (uint16_t)(s | 0x7c00u); return h; }        // >= 65520 rounds to infinity
    if (a < 0x38800000u) {                                                       // below 2^-14: subnormal half (or zero)
        // value * 2^24 rounded to an integer (ties to even): exact arithmetic in double
        const double scaled = (double)std::fabs(f) * 16777216.0;
        const double r = std::nearbyint(scaled);                                 // default rounding mode: to nearest even
        h.bits = (uint16_t)(s | (uint32_t)r);                                    // r == 1024 carries into the exponent: 2^-14
        return h;
    }
    const uint32_t e = (a >> 23) - 112u, m = a & 0x7fffffu;
    uint32_t hm = m >> 13;
    const uint32_t rest = m & 0x1fffu;
    uint32_t out = (e << 10) | hm;
    if (rest > 0x1000u || (rest == 0x1000u && (hm & 1u))) ++out;                 // carries propagate into the exponent
    h.bits = (uint16_t)(s | out);
    return h;
}

static inline float atomicAdd(float* p, float v) { const float old = *p; *p = old + v; return old; }
static inline int min(int a, int b) { return a < b ? a : b; }   // CUDA's device-side min(int, int)

// ---------------------------------------------------------------- torch::PackedTensorAccessor
namespace torch {
template <class T> struct RestrictPtrTraits { typedef T* PtrType; };
template <class T, size_t N, template <class> class PtrTraits = RestrictPtrTraits, class index_t = size_t>
struct PackedTensorAccessor {
    T* data;
    index_t sizes[N], strides[N];
    PackedTensorAccessor<T, N - 1, PtrTraits, index_t> operator[](index_t i) const
    {
        PackedTensorAccessor<T, N - 1, PtrTraits, index_t> sub;
        sub.data = data + i * strides[0];
        for (size_t k = 1; k < N; ++k) { sub.sizes[k - 1] = sizes[k]; sub.strides[k - 1] = strides[k]; }
        return sub;
    }
};
template <class T, template <class> class PtrTraits, class index_t>
struct PackedTensorAccessor<T, 1, PtrTraits, index_t> {
    T* data;
    index_t sizes[1], strides[1];
    T& operator[](index_t i) const { return data[i * strides[0]]; }
};
}  // namespace torch

// contiguous accessor over caller memory (the driver's helper, not part of any reference interface)
template <class T, size_t N>
static inline torch::PackedTensorAccessor<T, N, torch::RestrictPtrTraits, size_t> shim_accessor(const T* p, const size_t (&dims)[N])
{
    torch::PackedTensorAccessor<T, N, torch::RestrictPtrTraits, size_t> a;
    a.data = const_cast<T*>(p);
    size_t st = 1;
    for (size_t k = N; k-- > 0;) { a.sizes[k] = dims[k]; a.strides[k] = st; st *= dims[k]; }
    return a;
}

// ---------------------------------------------------------------- GLM (the subset ray_sampler.cu uses)
namespace glm {
struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
    float& operator[](int i) { return i == 0 ? x : y; }
    const float& operator[](int i) const { return i == 0 ? x : y; }
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const float& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct mat3 {
    vec3 col[3];   // column-major, 9 contiguous floats
    const vec3& operator[](int i) const { return col[i]; }
};
static_assert(sizeof(vec3) == 12 && sizeof(mat3) == 36, "GLM's packed layouts");
static inline vec3 operator/(float s, const vec3& v) { return vec3(s / v.x, s / v.y, s / v.z); }
static inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator+(const vec3& a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline float min(float x, float y) { return (y < x) ? y : x; }
static inline float max(float x, float y) { return (x < y) ? y : x; }
static inline vec3 min(const vec3& a, const vec3& b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
static inline vec3 max(const vec3& a, const vec3& b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
static inline float dot(const vec3& a, const vec3& b) { const vec3 t = a * b; return (t.x + t.y) + t.z; }
static inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
static inline vec3 normalize(const vec3& v) { return v * inversesqrt(dot(v, v)); }
// vec4 / mat4 (occupancy_grid_generation.cu): vec4(vec3, w); vec3 / float and vec3 - float component-wise; mat4 * vec4 in GLM's
// grouping (m[0] v.x + m[1] v.y) + (m[2] v.z + m[3] v.w), column vectors added component-wise
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};
struct mat4 {
    vec4 col[4];   // column-major, 16 contiguous floats
    const vec4& operator[](int i) const { return col[i]; }
};
static_assert(sizeof(vec4) == 16 && sizeof(mat4) == 64, "GLM's packed layouts");
static inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator-(const vec3& a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
static inline vec4 operator*(const vec4& a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline vec4 operator*(const mat4& m, const vec4& v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }
static inline vec3 operator*(const mat3& m, const vec3& v)
{
    return vec3((m[0][0] * v.x + m[1][0] * v.y) + m[2][0] * v.z, (m[0][1] * v.x + m[1][1] * v.y) + m[2][1] * v.z,
                (m[0][2] * v.x + m[1][2] * v.y) + m[2][2] * v.z);
}
}  // namespace glm
