// MFMA fragment helpers shared by the MLP kernels (mlp.hip, march.hip).
#pragma once
#include "hrf_common.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
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
// between layers, products accumulate in fp32 on the matrix cores: gfx950's v_mfma_f32_16x16x32_f16 / _bf16 wherever a
// contraction is 32 or more deep (every layer's forward and input-gradient product), the 16-deep
// v_mfma_f32_16x16x16_f16 / _bf16_1k for the weight-gradient products, which contract over the 16 samples of a tile. bf16 values
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
    // Two 16-deep blocks of a contraction in ONE gfx950 instruction (v_mfma_f32_16x16x32_f16): lane (g, c) supplies the
    // eight k-positions (g, 0..7) of A's row c and of B's column c; which logical k a position stands for is free as long
    // as A and B agree, so the two existing 4-element fragments of the blocks (k = 16 b + 4 g + r) are simply
    // concatenated -- the register chaining between layers (the C/D fragment of a layer is the B fragment of the next)
    // and the LDS weight layout stay as they are. Same products, same fp32 sum, half the matrix-core instructions.
    static __device__ __forceinline__ f4 mfma2(V a0, V a1, V b0, V b1, f4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                      __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7), c, 0, 0, 0);
    }
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
    static __device__ __forceinline__ f4 mfma2(V a0, V a1, V b0, V b1, f4 c)   // v_mfma_f32_16x16x32_bf16, see Prec<false>
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_shufflevector(__builtin_bit_cast(b4v, a0), __builtin_bit_cast(b4v, a1), 0, 1, 2, 3, 4, 5, 6, 7),
            __builtin_shufflevector(__builtin_bit_cast(b4v, b0), __builtin_bit_cast(b4v, b1), 0, 1, 2, 3, 4, 5, 6, 7), c, 0, 0, 0);
    }
    static __device__ __forceinline__ E from_f32(float x) { return hrf_f32_to_bf16(x); }
    static __device__ __forceinline__ float to_f32(E x) { return hrf_bf16_to_f32(x); }
    static __device__ __forceinline__ E from_half(_Float16 x) { return hrf_f32_to_bf16((float)x); }
    static __device__ __forceinline__ bool overflow(float v) { return !(fabsf(v) <= 3.3e38f); }   // bf16 has fp32's range
    static __device__ __forceinline__ V from_f4(f4 v) { return __builtin_bit_cast(V, __builtin_convertvector(v, b4v)); }
};
// acc += sum over NK 16-deep blocks of A_k . B_k, two blocks per instruction. An odd last block is paired with a zero
// B fragment rather than issued as a 16-deep instruction. Round 3 chose that because its first attempt at the mixed chain
// (16x16x16 accumulating onto a 16x16x32 result) failed every emb = 2 parity test. The instruction pair itself is NOT the
// cause: tools/microbench/mfma_chain_repro.hip runs exactly that chain on MI355X and gets the right sums (max error 3.7e-7,
// profiles/r04_microbench_mfma_chain_repro.txt), so the round-3 failure was in how that attempt filled its fragments. The
// padded form stays: the kernels that use it are bound by their dependent chains, not by MFMA issue (MFMA busy 41 % in
// k_mlp_bwd, profiles/r04_sq_k_mlp_bwd.txt), and it is the form every parity test has been run on.
template <class P> __device__ __forceinline__ typename P::V pv_zero();
template <class P, int NK, class FA, class FB>
__device__ __forceinline__ f4 contract(FA a, FB b, f4 acc)
{
#pragma unroll
    for (int k = 0; k + 1 < NK; k += 2) acc = P::mfma2(a(k), a(k + 1), b(k), b(k + 1), acc);
    if (NK & 1) acc = P::mfma2(a(NK - 1), a(NK - 1), b(NK - 1), pv_zero<P>(), acc);
    return acc;
}
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

// Copy a row-major (rows, cols) 16-bit matrix from global memory into LDS as rm[r*(cols+WPAD)+c] and / or as its transpose
// tr[c*(rows+WPAD)+r] (either destination may be NULL). 16-byte loads (8 consecutive elements of one row; cols % 16 == 0),
// all of a thread's loads independent of each other: NT = the workgroup size as a compile-time constant lets the loop unroll
// so that the loads of every matrix are in flight together -- the persistent backward kernel stages ten matrices before its
// first tile, and an element-wise loop with a wait per element cost it ~90 dependent memory round trips. NT = 0: blockDim.x.
template <int NT, class E16>
__device__ __forceinline__ void stage_rm_tr(E16* rm, E16* tr, const E16* __restrict__ src, int rows, int cols)
{
    const int nt = NT ? NT : (int)blockDim.x;
    const int chunks = (rows * cols) >> 3;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const int iters = (chunks + nt - 1) / nt;
#pragma unroll
        for (int k = 0; k < iters; ++k) {
            const int ch = k * nt + (int)threadIdx.x;
            if (ch < chunks) {
                const uint4 v = reinterpret_cast<const uint4*>(src)[ch];
                const int i = ch << 3, r = i / cols, c = i - r * cols;
                if (rm) {   // 8-byte aligned: (cols + WPAD) % 4 == 0 and c % 8 == 0
                    uint2* d = reinterpret_cast<uint2*>(rm + r * (cols + WPAD) + c);
                    d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
                }
                if (tr) {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint16_t e = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
                        tr[(c + j) * (rows + WPAD) + r] = __builtin_bit_cast(E16, e);
                    }
                }
            }
        }
    } else {   // unaligned source (a view at an odd offset): element-wise
        for (int i = threadIdx.x; i < rows * cols; i += nt) {
            const int r = i / cols, c = i - r * cols;
            const E16 e = src[i];
            if (rm) rm[r * (cols + WPAD) + c] = e;
            if (tr) tr[c * (rows + WPAD) + r] = e;
        }
    }
}
template <class E16>
__device__ __forceinline__ void stage_rm(E16* dst, const E16* __restrict__ src, int rows, int cols)
{
    stage_rm_tr<0, E16>(dst, (E16*)nullptr, src, rows, cols);
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

