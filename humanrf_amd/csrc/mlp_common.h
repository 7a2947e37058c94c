// MFMA fragment helpers shared by the MLP kernels (mlp.hip, march.hip).
#pragma once
#include "hrf_common.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma16(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 f4zero() { f4 z = {0.0f, 0.0f, 0.0f, 0.0f}; return z; }
__device__ __forceinline__ h4 to_h4(f4 v) { h4 r = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; return r; }
__device__ __forceinline__ h4 relu_h4(f4 v)
{
    h4 r = {(_Float16)fmaxf(v[0], 0.0f), (_Float16)fmaxf(v[1], 0.0f), (_Float16)fmaxf(v[2], 0.0f), (_Float16)fmaxf(v[3], 0.0f)};
    return r;
}
__device__ __forceinline__ float hround(float x) { return (float)(_Float16)x; }

// ---------------------------------------------------------------------------------------------
// Arithmetic type of the MLP kernels: fp16 (tcnn's FullyFusedMLP, the reference configuration) or bf16
// (BASELINE.json configs[4]: "fp16 hash tables + MFMA bf16 MLP"). Weights and activations are rounded to the 16-bit type
// between layers, products accumulate in fp32 on the matrix cores (v_mfma_f32_16x16x16_f16 / _bf16_1k). bf16 values
// are carried as their 16 bits in `short`s. Tensors that travel between kernels (features, h, rgb) stay fp16 containers
// in both modes: a bf16 value of moderate magnitude (2^-14 <= |x| <= 65504) is exactly representable in fp16.
// ---------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 b4v __attribute__((ext_vector_type(4)));

// round to nearest even; gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ short hrf_f32_to_bf16(float x) { return __builtin_bit_cast(short, (__bf16)x); }
__device__ __forceinline__ float hrf_bf16_to_f32(short b) { return __uint_as_float(((uint32_t)(uint16_t)b) << 16); }

template <bool kBF16> struct Prec;
template <> struct Prec<false> {
    typedef _Float16 E;
    typedef h4 V;
    static __device__ __forceinline__ f4 mfma(V a, V b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ E from_f32(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float to_f32(E x) { return (float)x; }
    static __device__ __forceinline__ E from_half(_Float16 x) { return x; }
    static __device__ __forceinline__ bool overflow(float v) { return !(fabsf(v) <= 65504.0f); }
    static __device__ __forceinline__ V from_f4(f4 v) { V r = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; return r; }
};
template <> struct Prec<true> {
    typedef short E;
    typedef s4v V;
    static __device__ __forceinline__ f4 mfma(V a, V b, f4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(b4v, a), __builtin_bit_cast(b4v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ E from_f32(float x) { return hrf_f32_to_bf16(x); }
    static __device__ __forceinline__ float to_f32(E x) { return hrf_bf16_to_f32(x); }
    static __device__ __forceinline__ E from_half(_Float16 x) { return hrf_f32_to_bf16((float)x); }
    static __device__ __forceinline__ bool overflow(float v) { return !(fabsf(v) <= 3.3e38f); }   // bf16 has fp32's range
    static __device__ __forceinline__ V from_f4(f4 v) { return __builtin_bit_cast(V, __builtin_convertvector(v, b4v)); }
};
template <class P> __device__ __forceinline__ typename P::V pv_from_f4(f4 v) { return P::from_f4(v); }
template <class P> __device__ __forceinline__ typename P::V pv_relu(f4 v)
{
    f4 r = {fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f), fmaxf(v[2], 0.0f), fmaxf(v[3], 0.0f)};
    return P::from_f4(r);
}
template <class P> __device__ __forceinline__ typename P::V pv_from_h4(h4 v)   // fp16 data in memory -> the kernel's type
{
    typename P::V r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = P::from_half(v[i]);
    return r;
}
template <class P> __device__ __forceinline__ typename P::V pv_zero()
{
    typename P::V r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = P::from_f32(0.0f);
    return r;
}
template <class P> __device__ __forceinline__ float p_round(float x) { return P::to_f32(P::from_f32(x)); }

#define WPAD 4  // halves of padding per LDS weight row (keeps 8-byte alignment, spreads banks)

// Copy a row-major (rows, cols) fp16 matrix from global memory into LDS as dst[r*(cols+WPAD)+c].
template <class E16>
__device__ __forceinline__ void stage_rm(E16* dst, const E16* src, int rows, int cols)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        dst[r * (cols + WPAD) + c] = src[i];
    }
}
// ... and its transpose dst[c*(rows+WPAD)+r].
template <class E16>
__device__ __forceinline__ void stage_tr(E16* dst, const E16* src, int rows, int cols)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        dst[c * (rows + WPAD) + r] = src[i];
    }
}
// A fragment of tile (rt, kt) of an LDS matrix with `cols` columns: lane (c = lane&15, g = lane>>4) reads
// M[16*rt + c][16*kt + 4g .. +3].
__device__ __forceinline__ h4 afrag(const _Float16* m, int cols, int rt, int kt, int lane)
{
    return *(const h4*)(m + (16 * rt + (lane & 15)) * (cols + WPAD) + 16 * kt + 4 * (lane >> 4));
}
__device__ __forceinline__ s4v afrag(const short* m, int cols, int rt, int kt, int lane)
{
    return *(const s4v*)(m + (16 * rt + (lane & 15)) * (cols + WPAD) + 16 * kt + 4 * (lane >> 4));
}

