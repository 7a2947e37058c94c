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

#define WPAD 4  // halves of padding per LDS weight row (keeps 8-byte alignment, spreads banks)

// Copy a row-major (rows, cols) fp16 matrix from global memory into LDS as dst[r*(cols+WPAD)+c].
__device__ __forceinline__ void stage_rm(_Float16* dst, const _Float16* src, int rows, int cols)
{
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        dst[r * (cols + WPAD) + c] = src[i];
    }
}
// ... and its transpose dst[c*(rows+WPAD)+r].
__device__ __forceinline__ void stage_tr(_Float16* dst, const _Float16* src, int rows, int cols)
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

