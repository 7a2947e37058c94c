// Fully fused density (sigma_net) and colour (color_net) MLPs for gfx950.
//
// Replaces tcnn FullyFusedMLP / Composite[SphericalHarmonics, Identity] as used at
// humanrf/scene_representation/humanrf.py:123-156,181-208 (+ truncated_exp, utils/activation.py:6-29).
// Semantics: SURVEY.md A.2/A.3 -- bias-free, fp16 weights and activations, ReLU, fp32 accumulation on the
// matrix cores (tcnn accumulates in half; DESIGN.md states the difference), outputs rounded to half.
//
// MI355X mapping: one wavefront owns a tile of 16 samples and keeps the whole layer chain in registers.
// All products are v_mfma_f32_16x16x16_f16 in the "transposed" form  H^T[hid][n] = W[hid][k] . X^T[k][n]:
// the C/D fragment of one layer (lane (g,c): rows 4g..4g+3, column c = sample) IS the B fragment of the
// next layer, so activations never touch LDS or HBM. Weights live in LDS (row-major and transposed copies,
// padded rows) and are read as 8-byte A fragments. In the backward kernel weight gradients contract over
// samples; the needed sample-major fragments come from one extra MFMA against an identity tile
// (D = X . I is an exact transpose of a fragment), so there is no LDS traffic for activations there either.
#include "hrf_common.h"

#include "mlp_common.h"
#include <stdlib.h>

// tcnn SphericalHarmonics degree 4 (A.3), component `i` of direction v in [-1,1]^3
__device__ __forceinline__ void sh16_all(float x, float y, float z, float* o)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * (x2 - y2);
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// The colour network's encoded input of a sample: columns [SH0..15 | geo0..G-1 | emb0..E-1 | ones] (A.3), G = geometry_feature_dim
// (15 in the reference's default architecture, model_args.py:22); lane group g supplies columns 16 kt + 4 g + j of tile kt.
// colour-network input width: 16 SH + G geometry features + E embedding dimensions, padded with ones to 16 KT columns; the kernels
// are instantiated for KT = 2 and 3
#define HRF_CHECK_COLOR_DIMS(G_, E_)                                                                                          \
    do {                                                                                                                      \
        HRF_CHECK_ARG((E_) >= 0 && (E_) <= 17, "camera_embedding_dim must be in [0,17]");                                      \
        HRF_CHECK_ARG((G_) >= 0 && (G_) <= 15, "geometry_feature_dim must be in [0,15]");                                      \
        HRF_CHECK_ARG((G_) + (E_) >= 1 && (G_) + (E_) <= 32, "16 + geometry_feature_dim + camera_embedding_dim must lie in (16, 48]"); \
    } while (0)
// n_hidden_layers_color (model_args.py:31): 1..3 hidden layers of 64 neurons; the NH - 1 hidden-to-hidden matrices arrive stacked
#define HRF_CHECK_COLOR_DEPTH(NH_, W2_)                                                                                       \
    do {                                                                                                                      \
        HRF_CHECK_ARG((NH_) >= 1 && (NH_) <= 3, "n_hidden_color (n_hidden_layers_color) must be 1, 2 or 3");                  \
        HRF_CHECK_ARG((NH_) == 1 || (W2_), "n_hidden_color > 1 needs the stacked hidden-to-hidden matrices");                 \
    } while (0)
#define HRF_DISPATCH_COLOR_DEPTH(NH_, M_)                                                                                     \
    do {                                                                                                                      \
        if ((NH_) == 1) M_(1); else if ((NH_) == 3) M_(3); else M_(2);                                                        \
    } while (0)
// ------------------------------------------------------------------------------------------------
// density forward: features (n,32) -> h (n,16) half, sigma = exp(h0) * density_scale
// ------------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void k_density_fwd(const _Float16* __restrict__ features, const typename P::E* __restrict__ w1,
                                                     const typename P::E* __restrict__ w2, float density_scale, int64_t n,
                                                     _Float16* __restrict__ out_h, float* __restrict__ out_sigma)
{
    typedef typename P::V V;
    __shared__ __attribute__((aligned(16))) typename P::E s_w1[64 * (32 + WPAD)];
    __shared__ __attribute__((aligned(16))) typename P::E s_w2[16 * (64 + WPAD)];
    stage_rm(s_w1, w1, 64, 32);
    stage_rm(s_w2, w2, 16, 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int64_t n_tiles = (n + 15) / 16;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    V a1[4][2], a2[4];
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
        a1[ht][0] = afrag(s_w1, 32, ht, 0, lane);
        a1[ht][1] = afrag(s_w1, 32, ht, 1, lane);
        a2[ht] = afrag(s_w2, 64, 0, ht, lane);
    }
    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t s = tile * 16 + c;
        h4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
        if (s < n) {
            x0 = *(const h4*)(features + s * 32 + 4 * g);
            x1 = *(const h4*)(features + s * 32 + 16 + 4 * g);
        }
        const V xv0 = pv_from_h4<P>(x0), xv1 = pv_from_h4<P>(x1);
        V hid[4];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) hid[ht] = pv_relu<P>(P::mfma2(a1[ht][0], a1[ht][1], xv0, xv1, f4zero()));
        f4 o = P::mfma2(a2[0], a2[1], hid[0], hid[1], f4zero());
        o = P::mfma2(a2[2], a2[3], hid[2], hid[3], o);
        if (s < n) {
            f4 orr;   // the network's output rounded to its 16-bit type, then carried in an fp16 container
#pragma unroll
            for (int r = 0; r < 4; ++r) orr[r] = p_round<P>(o[r]);
            const h4 oh = to_h4(orr);
            if (out_h) *(h4*)(out_h + s * 16 + 4 * g) = oh;
            if (out_sigma && g == 0) out_sigma[s] = expf((float)oh[0]) * density_scale;
        }
    }
}

extern "C" int hrf_density_mlp_fwd(const void* features, const void* w1, const void* w2, float density_scale,
                                   int64_t n, void* out_h, float* out_sigma, int mlp_bf16, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(features && w1 && w2, "NULL input");
    HRF_CHECK_ARG(out_h || out_sigma, "no output requested");
    const int64_t tiles = (n + 15) / 16;
    unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    if (mlp_bf16)
        hipLaunchKernelGGL(k_density_fwd<Prec<true>>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)features,
                           (const short*)w1, (const short*)w2, density_scale, n, (_Float16*)out_h, out_sigma);
    else
        hipLaunchKernelGGL(k_density_fwd<Prec<false>>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)features,
                           (const _Float16*)w1, (const _Float16*)w2, density_scale, n, (_Float16*)out_h, out_sigma);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// colour forward: (dir[ray], geo = h[1:16], camera embedding) -> rgb (n,3) half
// ------------------------------------------------------------------------------------------------
// NH = n_hidden_layers_color (model_args.py:31; 1..3): w2 holds the NH - 1 hidden-to-hidden matrices one after the other, the order
// tcnn's FullyFusedMLP keeps them in its flat parameter vector
// (Bounded to four wavefronts per SIMD up to two hidden layers: the generic form came out at 121 + 8 accumulation registers, one over
// the 128 that four wavefronts leave each, and ran 45 -> 57 us.)
template <int KT, class P, int NH = 2>
__global__ __launch_bounds__(256, (NH <= 2 ? 4 : 2)) void k_color_fwd(
    const float* __restrict__ ray_dirs, const int64_t* __restrict__ sample_ray, const _Float16* __restrict__ h,
    const float* __restrict__ cam_emb, const int32_t* __restrict__ ray_cameras, int E, int use_emb,
    const typename P::E* __restrict__ w1, const typename P::E* __restrict__ w2, const typename P::E* __restrict__ w3, int64_t n,
    _Float16* __restrict__ out_rgb, int G)
{
    typedef typename P::V V;
    constexpr int KIN = 16 * KT;
    constexpr int NMID = NH - 1, W2SZ = 64 * (64 + WPAD);
    __shared__ __attribute__((aligned(16))) typename P::E s_w1[64 * (KIN + WPAD)];
    __shared__ __attribute__((aligned(16))) typename P::E s_w2[(NMID > 0 ? NMID : 1) * W2SZ];
    __shared__ __attribute__((aligned(16))) typename P::E s_w3[16 * (64 + WPAD)];
    stage_rm(s_w1, w1, 64, KIN);
#pragma unroll
    for (int m = 0; m < NMID; ++m) stage_rm(s_w2 + m * W2SZ, w2 + m * 4096, 64, 64);
    stage_rm(s_w3, w3, 16, 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int64_t n_tiles = (n + 15) / 16;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t s = tile * 16 + c;
        V x[KT];
        {
            // Lane (g, c) supplies columns 16 kt + 4 g + j of sample c: its four SH components and, per further tile, four of the
            // [geo | embedding | ones] columns -- loaded where they are needed. (Rounds 1-5 filled sh[16], geo[15], emb[16] per lane
            // and indexed them with 4 g + j: 80 bytes of scratch per lane and 134 registers, three wavefronts per SIMD.)
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            int cam = 0;
            const bool live = s < n;
            if (live) {
                const int64_t r = sample_ray[s];
                // humanrf.py:192 maps directions to [0,1]; tcnn's SH maps them back with 2x-1
                dx = ((ray_dirs[r * 3 + 0] + 1.0f) * 0.5f) * 2.0f - 1.0f;
                dy = ((ray_dirs[r * 3 + 1] + 1.0f) * 0.5f) * 2.0f - 1.0f;
                dz = ((ray_dirs[r * 3 + 2] + 1.0f) * 0.5f) * 2.0f - 1.0f;
                if (E > 0 && use_emb) cam = ray_cameras[r];
            }
            float sh[16];
            sh16_all(dx, dy, dz, sh);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = g == 0 ? sh[j] : g == 1 ? sh[4 + j] : g == 2 ? sh[8 + j] : sh[12 + j];
                x[0][j] = P::from_f32(v);
            }
#pragma unroll
            for (int kt = 1; kt < KT; ++kt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ii = 16 * (kt - 1) + 4 * g + j;       // column 16 + ii: geo[ii], then the embedding, then ones
                    float v = 1.0f;
                    if (ii < G) v = live ? (float)h[s * 16 + 1 + ii] : 0.0f;
                    else if (ii < G + E) v = (live && use_emb) ? cam_emb[cam * E + (ii - G)] : 0.0f;
                    x[kt][j] = P::from_f32(v);
                }
            }
        }
        V hh[NH][4];                       // hidden activations, layer by layer
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            f4 acc = f4zero();
            acc = contract<P, KT>([&](int kt) { return afrag(s_w1, KIN, ht, kt, lane); }, [&](int kt) { return x[kt]; }, acc);
            hh[0][ht] = pv_relu<P>(acc);
        }
#pragma unroll
        for (int m = 0; m < NMID; ++m) {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                f4 acc = f4zero();
                acc = contract<P, 4>([&](int kt) { return afrag(s_w2 + m * W2SZ, 64, ht, kt, lane); },
                                     [&](int kt) { return hh[m][kt]; }, acc);
                hh[m + 1][ht] = pv_relu<P>(acc);
            }
        }
        f4 o = f4zero();
        o = contract<P, 4>([&](int kt) { return afrag(s_w3, 64, 0, kt, lane); }, [&](int kt) { return hh[NH - 1][kt]; }, o);
        if (s < n && g == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) out_rgb[s * 3 + k] = (_Float16)p_round<P>(1.0f / (1.0f + expf(-o[k])));
        }
    }
}

extern "C" int hrf_color_mlp_fwd(const float* ray_dirs, const int64_t* sample_ray, const void* h,
                                 const float* cam_emb, const int32_t* ray_cameras, int emb_dim, int use_emb,
                                 const void* w1, const void* w2, const void* w3, int64_t n, void* out_rgb,
                                 int mlp_bf16, int geometry_feature_dim, int n_hidden_color, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_COLOR_DEPTH(n_hidden_color, w2);
    HRF_CHECK_ARG(ray_dirs && sample_ray && h && w1 && w3 && out_rgb, "NULL argument");
    HRF_CHECK_COLOR_DIMS(geometry_feature_dim, emb_dim);
    const int G = geometry_feature_dim;
    HRF_CHECK_ARG(!(use_emb && emb_dim > 0) || (cam_emb && ray_cameras), "embedding requested without table");
    const int64_t tiles = (n + 15) / 16;
    unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    const int KT = (16 + G + emb_dim + 15) / 16;
#define HRF_LAUNCH_CF(K, PP, ET, NHC)                                                                                \
    hipLaunchKernelGGL((k_color_fwd<K, PP, NHC>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, ray_dirs, sample_ray,  \
                       (const _Float16*)h, cam_emb, ray_cameras, emb_dim, use_emb, (const ET*)w1, (const ET*)w2,     \
                       (const ET*)w3, n, (_Float16*)out_rgb, G)
#define HRF_LAUNCH_CF_P(NHC)                                                                                          \
    do {                                                                                                              \
        if (mlp_bf16) { if (KT == 2) HRF_LAUNCH_CF(2, Prec<true>, short, NHC); else HRF_LAUNCH_CF(3, Prec<true>, short, NHC); } \
        else { if (KT == 2) HRF_LAUNCH_CF(2, Prec<false>, _Float16, NHC); else HRF_LAUNCH_CF(3, Prec<false>, _Float16, NHC); }  \
    } while (0)
    HRF_DISPATCH_COLOR_DEPTH(n_hidden_color, HRF_LAUNCH_CF_P);
#undef HRF_LAUNCH_CF_P
#undef HRF_LAUNCH_CF
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward of both networks (activations recomputed from the encoded features)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ h4 to_h4_chk(f4 v, bool& bad)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) bad |= !(fabsf(v[i]) <= 65504.0f);
    return to_h4(v);
}
template <class P>
__device__ __forceinline__ typename P::V pv_chk(f4 v, bool& bad)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) bad |= P::overflow(v[i]);
    return pv_from_f4<P>(v);
}
template <class P>
__device__ __forceinline__ typename P::V relu_mask(f4 d, typename P::V act, bool& bad)
{
    f4 m;
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = (P::to_f32(act[i]) > 0.0f) ? d[i] : 0.0f;
    return pv_chk<P>(m, bad);
}
// exact transpose of a 16x16 fragment through the matrix core: D = X . I
template <class P>
__device__ __forceinline__ typename P::V transpose_frag(typename P::V t, typename P::V ident)
{
    return pv_from_f4<P>(P::mfma(t, ident, f4zero()));
}

// Features of one 16-sample tile of k_mlp_bwd, as loaded from memory (lane (g,c) holds sample c of the tile).
struct MbTileIn {
    h4 xa, xb;           // features 4g..4g+3 and 16+4g..
};
__device__ __forceinline__ int64_t mb_load_ray(const int64_t* __restrict__ sample_ray, int64_t tile, int c, int64_t n)
{
    const int64_t s = tile * 16 + c;
    return s < n ? sample_ray[s] : (int64_t)0;
}
__device__ __forceinline__ MbTileIn mb_load_tile(const _Float16* __restrict__ features, int64_t tile, int g, int c, int64_t n)
{
    MbTileIn in;
    in.xa = h4{0, 0, 0, 0}; in.xb = h4{0, 0, 0, 0};
    const int64_t s = tile * 16 + c;
    if (s < n) {
        in.xa = *(const h4*)(features + s * 32 + 4 * g);
        in.xb = *(const h4*)(features + s * 32 + 16 + 4 * g);
    }
    return in;
}

// (Measured, round 2: software-pipelining these loads one tile ahead changes nothing -- 0.29 ms with and without; with one
// wavefront per SIMD the kernel waits on the LDS fragment read in front of every MFMA, not on global memory.)
// MODE 0: both networks of the field (the training step); 1: sigma_net alone, upstream gradient d_h (n,16) fp32 given
// directly (d_features out as in mode 0); 2: color_net alone, its geometry input read from h_in (n,16) half, the gradient
// with respect to that input written to d_features as (n,16) fp32 (row 0, the density logit, is zero). Modes 1 and 2 are
// the backward passes of the stand-alone tcnn-shaped modules (humanrf_amd.compat.tinycudann.Network /
// NetworkWithInputEncoding, humanrf.py:123-156).
// Registers: the fused form (MODE 0) holds 176 weight-gradient accumulator registers + the parked weight fragments and runs
// one wavefront per SIMD; the single-network forms are bounded to two wavefronts per SIMD (256 registers).
// NH: hidden layers of the colour network (1..3, k_color_fwd); cw2 / g_cw2 hold the NH - 1 hidden-to-hidden matrices stacked. With
// three hidden layers the colour-alone form carries 128 more accumulator registers and runs one wavefront per SIMD as well.
template <int KT, class P, int MODE = 0, int NH = 2>
__global__ __launch_bounds__(256, ((MODE == 0 || NH > 2) ? 1 : 2)) void k_mlp_bwd(
    const _Float16* __restrict__ features, const float* __restrict__ ray_dirs, const int64_t* __restrict__ sample_ray,
    const float* __restrict__ cam_emb, const int32_t* __restrict__ ray_cameras, int E, int use_emb,
    const typename P::E* __restrict__ sw1, const typename P::E* __restrict__ sw2, const typename P::E* __restrict__ cw1,
    const typename P::E* __restrict__ cw2, const typename P::E* __restrict__ cw3, float density_scale,
    const float* __restrict__ d_rgb, const float* __restrict__ d_sigma, int64_t n, void* __restrict__ d_features, int df_fp32,
    float* __restrict__ g_sw1, float* __restrict__ g_sw2, float* __restrict__ g_cw1, float* __restrict__ g_cw2,
    float* __restrict__ g_cw3, float* __restrict__ g_emb, int32_t* __restrict__ flags,
    const float* __restrict__ d_h = nullptr, const _Float16* __restrict__ h_in = nullptr, float gb = 0.0f, int G = 15)
{
    // gb > 0 (hrf_mlp_bwd's grad_boundary): the two places where the reference's gradient is a HALF tensor at the
    // GradScaler's scale between tcnn modules -- dL/d(sigma_net output) and dL/d(features) -- round through half at 1 / gb of
    // the fused scale (hrf_through_half); inside a network tcnn's backward runs in half at the fused scale, as this kernel does
    const float inv_gb = gb > 0.0f ? 1.0f / gb : 0.0f;
    typedef typename P::V V;
    typedef typename P::E EW;
    constexpr int KIN = 16 * KT;
    constexpr bool SIGMA = MODE != 2, COLOR = MODE != 1;
    constexpr int NMID = NH - 1, NMID1 = NMID > 0 ? NMID : 1, W2SZ = 64 * (64 + WPAD);
    // forward (row-major) and transposed copies of all five weight matrices
    __shared__ __attribute__((aligned(16))) EW s_sw1[64 * (32 + WPAD)], s_sw1t[32 * (64 + WPAD)];
    __shared__ __attribute__((aligned(16))) EW s_sw2[16 * (64 + WPAD)], s_sw2t[64 * (16 + WPAD)];
    __shared__ __attribute__((aligned(16))) EW s_cw1[64 * (KIN + WPAD)], s_cw1t[KIN * (64 + WPAD)];
    __shared__ __attribute__((aligned(16))) EW s_cw2[NMID1 * W2SZ], s_cw2t[NMID1 * W2SZ];
    __shared__ __attribute__((aligned(16))) EW s_cw3[16 * (64 + WPAD)], s_cw3t[64 * (16 + WPAD)];
    if constexpr (SIGMA) {
        stage_rm_tr<256>(s_sw1, s_sw1t, sw1, 64, 32);
        stage_rm_tr<256>(s_sw2, s_sw2t, sw2, 16, 64);
    }
    if constexpr (COLOR) {
        stage_rm_tr<256>(s_cw1, s_cw1t, cw1, 64, KIN);
#pragma unroll
        for (int m = 0; m < NMID; ++m) stage_rm_tr<256>(s_cw2 + m * W2SZ, s_cw2t + m * W2SZ, cw2 + m * 4096, 64, 64);
        stage_rm_tr<256>(s_cw3, s_cw3t, cw3, 16, 64);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int64_t n_tiles = (n + 15) / 16;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    V ident;
#pragma unroll
    for (int j = 0; j < 4; ++j) ident[j] = P::from_f32((4 * g + j == c) ? 1.0f : 0.0f);

    // weight-gradient accumulators, fragment (ot, it): lane (g,c) holds dW[16*ot + 4g + r][16*it + c]
    f4 acc_sw1[4][2], acc_sw2[4], acc_cw1[4][KT], acc_cw2[NMID1][4][4], acc_cw3[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        acc_sw2[a] = f4zero(); acc_cw3[a] = f4zero();
#pragma unroll
        for (int b = 0; b < 2; ++b) acc_sw1[a][b] = f4zero();
#pragma unroll
        for (int b = 0; b < KT; ++b) acc_cw1[a][b] = f4zero();
#pragma unroll
        for (int m = 0; m < NMID1; ++m)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc_cw2[m][a][b] = f4zero();
    }
    bool bad = false;

    const bool want_cam = E > 0 && use_emb;
    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        // single-network forms: the weight fragments are read from LDS where they are used. (Left alone, the compiler hoists
        // every fragment out of this loop and parks it in registers: ~250 of them, which is what holds the fused form to
        // one wavefront per SIMD; at two per SIMD the LDS reads hide behind the other wavefront.)
        if constexpr (MODE != 0) asm volatile("" ::: "memory");
        const int64_t s = tile * 16 + c;
        const bool valid = s < n;
        MbTileIn in;
        in.xa = h4{0, 0, 0, 0}; in.xb = h4{0, 0, 0, 0};
        if constexpr (SIGMA) in = mb_load_tile(features, tile, g, c, n);
        float dir0 = -1.0f, dir1 = -1.0f, dir2 = -1.0f;
        int cam_in = 0;
        if constexpr (COLOR) {
            const int64_t ray = mb_load_ray(sample_ray, tile, c, n);
            if (valid) {
                dir0 = ray_dirs[ray * 3 + 0]; dir1 = ray_dirs[ray * 3 + 1]; dir2 = ray_dirs[ray * 3 + 2];
                if (want_cam) cam_in = ray_cameras[ray];
            }
        }
        // upstream gradients of this tile: needed after the forward recompute, which hides their latency
        float up_rgb[3] = {0.0f, 0.0f, 0.0f}, up_sigma = 0.0f;
        if (valid && g == 0) {
            if constexpr (COLOR) { up_rgb[0] = d_rgb[s * 3 + 0]; up_rgb[1] = d_rgb[s * 3 + 1]; up_rgb[2] = d_rgb[s * 3 + 2]; }
            if constexpr (MODE == 0) up_sigma = d_sigma[s];
            if constexpr (MODE == 2) { if (d_sigma) up_sigma = d_sigma[s]; }
        }
        f4 up_h = f4zero();                       // MODE 1: the upstream gradient of sigma_net's 16 outputs, rows 4g..4g+3
        if constexpr (MODE == 1) { if (valid) up_h = *(const f4*)(d_h + s * 16 + 4 * g); }
        // ---------------- forward recompute ----------------
        V xf[2];
        xf[0] = pv_from_h4<P>(in.xa); xf[1] = pv_from_h4<P>(in.xb);
        V hs[4];
        f4 ho = f4zero();
        if constexpr (SIGMA) {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                hs[ht] = pv_relu<P>(P::mfma2(afrag(s_sw1, 32, ht, 0, lane), afrag(s_sw1, 32, ht, 1, lane), xf[0], xf[1], f4zero()));
            }
            ho = contract<P, 4>([&](int kt) { return afrag(s_sw2, 64, 0, kt, lane); }, [&](int kt) { return hs[kt]; }, ho);
        } else {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) hs[ht] = pv_zero<P>();
        }
        // sigma_net output is a half tensor: lane (g,c) holds h[4g + r] of sample c
        float hof[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hof[r] = p_round<P>(ho[r]);
        if constexpr (MODE == 2) {                // the geometry input of the colour network comes from the caller
            const h4 hv = valid ? *(const h4*)(h_in + s * 16 + 4 * g) : h4{0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) hof[r] = (float)hv[r];
        }
        // geometry features geo[i] = h[1 + i]; this lane needs geo[4g + j] = h[4g + j + 1], j = 0..3
        float geo_l[4];
        {
            const float nxt = __shfl(hof[0], (lane + 16) & 63, 64);  // h[4(g+1)] from lane group g+1
            geo_l[0] = hof[1]; geo_l[1] = hof[2]; geo_l[2] = hof[3]; geo_l[3] = nxt;
        }
        f4 dx01 = f4zero();     // d/d(colour-network input columns 16 + 4g + r) of sample c: the geometry features
        if constexpr (COLOR) {
        int cam = 0;
        V x0[KT];
        {
            float sh[16];
            float dx = 0.0f, dy = 0.0f, dz = 0.0f;
            if (valid) {
                dx = ((dir0 + 1.0f) * 0.5f) * 2.0f - 1.0f;
                dy = ((dir1 + 1.0f) * 0.5f) * 2.0f - 1.0f;
                dz = ((dir2 + 1.0f) * 0.5f) * 2.0f - 1.0f;
                cam = cam_in;
            }
            sh16_all(dx, dy, dz, sh);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 16 * kt + 4 * g + j;
                    float v;
                    if (kt == 0) v = sh[4 * g + j];
                    else {
                        const int ii = col - 16;
                        if (ii < G) v = geo_l[j];   // only reachable for kt == 1 (G <= 15): ii = 4g + j
                        else if (ii < G + E) v = (use_emb && valid) ? cam_emb[cam * E + (ii - G)] : 0.0f;
                        else v = 1.0f;
                    }
                    x0[kt][j] = P::from_f32(v);
                }
            }
        }
        V hh[NH][4];                        // hidden activations of the colour network, layer by layer
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            f4 acc = f4zero();
            acc = contract<P, KT>([&](int kt) { return afrag(s_cw1, KIN, ht, kt, lane); }, [&](int kt) { return x0[kt]; }, acc);
            hh[0][ht] = pv_relu<P>(acc);
        }
#pragma unroll
        for (int m = 0; m < NMID; ++m) {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
                f4 acc = f4zero();
                acc = contract<P, 4>([&](int kt) { return afrag(s_cw2 + m * W2SZ, 64, ht, kt, lane); },
                                     [&](int kt) { return hh[m][kt]; }, acc);
                hh[m + 1][ht] = pv_relu<P>(acc);
            }
        }
        f4 o = f4zero();
        o = contract<P, 4>([&](int kt) { return afrag(s_cw3, 64, 0, kt, lane); }, [&](int kt) { return hh[NH - 1][kt]; }, o);

        // ---------------- backward ----------------
        // dO[o][n]: rows 0..2 carry d_rgb * sigmoid'(z)
        f4 dO = f4zero();
        if (valid && g == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float sg = 1.0f / (1.0f + expf(-o[k]));
                dO[k] = up_rgb[k] * (sg * (1.0f - sg));
            }
        }
        const V dOh = pv_chk<P>(dO, bad);
        const V dO_nt = transpose_frag<P>(dOh, ident);
        // output layer: dW3 += dO^T-frag x H_last ; dH_last = W3^T dO
        V dh1[4];                           // gradient of the current hidden layer's activations, walking down to the first
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc_cw3[t] = P::mfma(dO_nt, transpose_frag<P>(hh[NH - 1][t], ident), acc_cw3[t]);
            dh1[t] = relu_mask<P>(P::mfma(afrag(s_cw3t, 16, t, 0, lane), dOh, f4zero()), hh[NH - 1][t], bad);
        }
        // hidden-to-hidden layers, last to first
#pragma unroll
        for (int m = NMID - 1; m >= 0; --m) {
            V in_nt[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) in_nt[t] = transpose_frag<P>(hh[m][t], ident);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                const V d_nt = transpose_frag<P>(dh1[ot], ident);
#pragma unroll
                for (int it = 0; it < 4; ++it) acc_cw2[m][ot][it] = P::mfma(d_nt, in_nt[it], acc_cw2[m][ot][it]);
            }
            V dnext[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f4 acc = f4zero();
                acc = contract<P, 4>([&](int kt) { return afrag(s_cw2t + m * W2SZ, 64, t, kt, lane); },
                                     [&](int kt) { return dh1[kt]; }, acc);
                dnext[t] = relu_mask<P>(acc, hh[m][t], bad);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) dh1[t] = dnext[t];
        }
        // colour layer 1: dW1 and the input gradient of the identity part (geo, embedding)
        V x0_nt[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) x0_nt[kt] = transpose_frag<P>(x0[kt], ident);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const V d_nt = transpose_frag<P>(dh1[ot], ident);
#pragma unroll
            for (int it = 0; it < KT; ++it) acc_cw1[ot][it] = P::mfma(d_nt, x0_nt[it], acc_cw1[ot][it]);
        }
        f4 dx0[KT];  // dx0[kt]: lane (g,c) holds d/d(input col 16kt + 4g + r) of sample c; tile 0 (SH) not needed
#pragma unroll
        for (int kt = 1; kt < KT; ++kt) {
            f4 acc = f4zero();
            acc = contract<P, 4>([&](int ht) { return afrag(s_cw1t, 64, kt, ht, lane); }, [&](int ht) { return dh1[ht]; }, acc);
            dx0[kt] = acc;
        }
        // camera embedding gradient: input columns 16+G .. 16+G+E-1
        if (E > 0 && use_emb) {
#pragma unroll
            for (int kt = 1; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ii = 16 * kt + 4 * g + r - 16;
                    const bool is_emb = valid && ii >= G && ii < G + E;
                    // all samples of a ray share the camera: aggregate equal keys in the wave first
                    unsigned long long todo = __ballot(is_emb);
                    const uint32_t key = (uint32_t)(cam * E + (ii - G));
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
                        const bool mine = is_emb && key == k0;
                        const unsigned long long m = __ballot(mine);
                        float v = mine ? dx0[kt][r] : 0.0f;
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                        if (lane == leader) unsafeAtomicAdd(g_emb + k0, v);
                        todo &= ~m;
                    }
                }
            }
        }
        dx01 = dx0[1];
        }
        // d h[o][n], o = 4g + r: o = 0 from sigma (truncated_exp backward), o >= 1 from geo = input col 15 + o
        f4 dho;
        if constexpr (MODE == 1) {
            dho = up_h;
        } else {
            // dx01 holds columns 16 + 4g + r  <->  h index 4g + r + 1; shift down by one row
            const float prev = __shfl(dx01[3], (lane + 48) & 63, 64);  // row 4(g-1)+3 from lane group g-1
            dho[1] = dx01[0]; dho[2] = dx01[1]; dho[3] = dx01[2];
            dho[0] = prev;
            if (g == 0) {
                float ds = 0.0f;
                if constexpr (MODE == 0 || MODE == 2) {     // (MODE 2: only when the caller hands d_sigma over, up_sigma = 0 otherwise)
                    if (valid) ds = up_sigma * (density_scale * expf(fminf(fmaxf(hof[0], -15.0f), 15.0f)));
                }
                dho[0] = ds;
            }
            // sigma_net outputs beyond the G geometry features feed nothing (with G < 15 the columns behind the geometry block are
            // embedding / padding columns, whose input gradient is not a gradient of h)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r > G) dho[r] = 0.0f;
            if (!valid) dho = f4zero();
        }
        if constexpr (MODE == 2) {                // colour network alone: the gradient of its geometry input is the result
            if (valid) *(f4*)((float*)d_features + s * 16 + 4 * g) = dho;
            continue;
        }
        if (gb > 0.0f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dho[r] = hrf_through_half(dho[r], gb, inv_gb);
                bad |= !(fabsf(dho[r]) < 3.0e38f);      // |x / gb| > 65504: the reference's half tensor holds inf there
            }
        }
        const V dhoh = pv_chk<P>(dho, bad);
        const V dho_nt = transpose_frag<P>(dhoh, ident);
        // sigma layer 2
        V dhs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc_sw2[t] = P::mfma(dho_nt, transpose_frag<P>(hs[t], ident), acc_sw2[t]);
            dhs[t] = relu_mask<P>(P::mfma(afrag(s_sw2t, 16, t, 0, lane), dhoh, f4zero()), hs[t], bad);
        }
        // sigma layer 1
        const V xf_nt0 = transpose_frag<P>(xf[0], ident), xf_nt1 = transpose_frag<P>(xf[1], ident);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const V d_nt = transpose_frag<P>(dhs[ot], ident);
            acc_sw1[ot][0] = P::mfma(d_nt, xf_nt0, acc_sw1[ot][0]);
            acc_sw1[ot][1] = P::mfma(d_nt, xf_nt1, acc_sw1[ot][1]);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f4 acc = f4zero();
            acc = contract<P, 4>([&](int ht) { return afrag(s_sw1t, 64, kt, ht, lane); }, [&](int ht) { return dhs[ht]; }, acc);
            if (gb > 0.0f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r] = hrf_through_half(acc[r], gb, inv_gb);
                    bad |= !(fabsf(acc[r]) < 3.0e38f);  // a finite fp32 value the half boundary turns into inf: found_inf, like the reference
                }
            }
            if (df_fp32 == 2) {
                // level-major fp32: dY_lm[level][sample] = (f[2*level], f[2*level+1]); this lane holds features
                // 16kt + 4g .. +3 = levels 8kt + 2g and 8kt + 2g + 1 of sample s
                if (valid) {
                    float2* lm = (float2*)d_features;
                    lm[(size_t)(8 * kt + 2 * g) * n + s] = make_float2(acc[0], acc[1]);
                    lm[(size_t)(8 * kt + 2 * g + 1) * n + s] = make_float2(acc[2], acc[3]);
                }
            } else if (df_fp32 == 1) {
                if (valid) *(f4*)((float*)d_features + s * 32 + 16 * kt + 4 * g) = acc;
            } else {
                const h4 df = to_h4_chk(acc, bad);
                if (valid) *(h4*)((_Float16*)d_features + s * 32 + 16 * kt + 4 * g) = df;
            }
        }
    }

    // flush the weight-gradient fragments
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ot + 4 * g + r;
            if constexpr (SIGMA) {
#pragma unroll
                for (int it = 0; it < 2; ++it) unsafeAtomicAdd(g_sw1 + row * 32 + 16 * it + c, acc_sw1[ot][it][r]);
            }
            if constexpr (COLOR) {
#pragma unroll
                for (int it = 0; it < KT; ++it) unsafeAtomicAdd(g_cw1 + row * KIN + 16 * it + c, acc_cw1[ot][it][r]);
#pragma unroll
                for (int m = 0; m < NMID; ++m)
#pragma unroll
                    for (int it = 0; it < 4; ++it) unsafeAtomicAdd(g_cw2 + m * 4096 + row * 64 + 16 * it + c, acc_cw2[m][ot][it][r]);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (SIGMA) unsafeAtomicAdd(g_sw2 + (4 * g + r) * 64 + 16 * it + c, acc_sw2[it][r]);
            if constexpr (COLOR) unsafeAtomicAdd(g_cw3 + (4 * g + r) * 64 + 16 * it + c, acc_cw3[it][r]);
        }
    }
    if (__any(bad) && lane == 0) atomicOr(flags, 1);
}

extern "C" int hrf_mlp_bwd(const void* features, const float* ray_dirs, const int64_t* sample_ray,
                           const float* cam_emb, const int32_t* ray_cameras, int emb_dim, int use_emb,
                           const void* sw1, const void* sw2, const void* cw1, const void* cw2, const void* cw3,
                           float density_scale, const float* d_rgb, const float* d_sigma, int64_t n,
                           void* d_features, int d_features_fp32, float grad_boundary, float* d_sw1, float* d_sw2,
                           float* d_cw1, float* d_cw2, float* d_cw3, float* d_cam_emb, int32_t* flags, int mlp_bf16,
                           int geometry_feature_dim, int n_hidden_color, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_COLOR_DEPTH(n_hidden_color, cw2 && d_cw2);
    HRF_CHECK_ARG(grad_boundary >= 0.0f, "grad_boundary must be 0 (off) or the factor between the fused and the reference's gradient scale");
    HRF_CHECK_ARG(features && ray_dirs && sample_ray && sw1 && sw2 && cw1 && cw3, "NULL input");
    HRF_CHECK_ARG(d_rgb && d_sigma && d_features && d_sw1 && d_sw2 && d_cw1 && d_cw3 && flags, "NULL gradient buffer");
    HRF_CHECK_COLOR_DIMS(geometry_feature_dim, emb_dim);
    const int G = geometry_feature_dim;
    HRF_CHECK_ARG(!(use_emb && emb_dim > 0) || (cam_emb && ray_cameras && d_cam_emb), "embedding requested without table");
    const int64_t tiles = (n + 15) / 16;
    unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (blocks > 256) blocks = 256;  // persistent: one workgroup per CU, accumulators flushed once per wave
    const int KT = (16 + G + emb_dim + 15) / 16;
#define HRF_LAUNCH_MB(K, PP, ET, NHC)                                                                                 \
    hipLaunchKernelGGL((k_mlp_bwd<K, PP, 0, NHC>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)features, \
                       ray_dirs, sample_ray, cam_emb, ray_cameras, emb_dim, (use_emb && emb_dim > 0) ? 1 : 0,         \
                       (const ET*)sw1, (const ET*)sw2, (const ET*)cw1, (const ET*)cw2, (const ET*)cw3, density_scale,  \
                       d_rgb, d_sigma, n, d_features, d_features_fp32, d_sw1, d_sw2, d_cw1, d_cw2, d_cw3, d_cam_emb, flags,         \
                       (const float*)nullptr, (const _Float16*)nullptr, grad_boundary, G)
#define HRF_LAUNCH_MB_P(NHC)                                                                                          \
    do {                                                                                                              \
        if (mlp_bf16) { if (KT == 2) HRF_LAUNCH_MB(2, Prec<true>, short, NHC); else HRF_LAUNCH_MB(3, Prec<true>, short, NHC); } \
        else { if (KT == 2) HRF_LAUNCH_MB(2, Prec<false>, _Float16, NHC); else HRF_LAUNCH_MB(3, Prec<false>, _Float16, NHC); }  \
    } while (0)
    HRF_DISPATCH_COLOR_DEPTH(n_hidden_color, HRF_LAUNCH_MB_P);
#undef HRF_LAUNCH_MB_P
#undef HRF_LAUNCH_MB
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The two networks differentiated separately: the backward passes of tcnn.Network (sigma_net) and
// tcnn.NetworkWithInputEncoding (color_net) as stand-alone modules (humanrf.py:123-156), for code written against
// tinycudann's module surface (humanrf_amd.compat.tinycudann). Same kernel as hrf_mlp_bwd with one network compiled out.
// ------------------------------------------------------------------------------------------------
extern "C" int hrf_density_mlp_bwd(const void* features, const void* w1, const void* w2, const float* d_h, int64_t n,
                                   void* d_features, int d_features_fp32, float grad_boundary, float* d_w1, float* d_w2,
                                   int32_t* flags, int mlp_bf16, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(features && w1 && w2 && d_h && d_features && d_w1 && d_w2 && flags, "NULL argument");
    HRF_CHECK_ARG(d_features_fp32 >= 0 && d_features_fp32 <= 2, "d_features_fp32 must be 0 (fp16), 1 (fp32) or 2 (fp32 level-major)");
    HRF_CHECK_ARG(grad_boundary >= 0.0f, "grad_boundary must be >= 0");
    const int64_t tiles = (n + 15) / 16;
    unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (blocks > 768) blocks = 768;      // persistent: three workgroups per CU (136 registers)
#define HRF_LAUNCH_DB(PP, ET)                                                                                          \
    hipLaunchKernelGGL((k_mlp_bwd<2, PP, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)features, \
                       (const float*)nullptr, (const int64_t*)nullptr, (const float*)nullptr, (const int32_t*)nullptr, 0, 0, \
                       (const ET*)w1, (const ET*)w2, (const ET*)nullptr, (const ET*)nullptr, (const ET*)nullptr, 1.0f,     \
                       (const float*)nullptr, (const float*)nullptr, n, d_features, d_features_fp32, d_w1, d_w2,            \
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, flags, d_h, (const _Float16*)nullptr,       \
                       grad_boundary)
    if (mlp_bf16) HRF_LAUNCH_DB(Prec<true>, short); else HRF_LAUNCH_DB(Prec<false>, _Float16);
#undef HRF_LAUNCH_DB
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_color_mlp_bwd(const float* ray_dirs, const int64_t* sample_ray, const void* h, const float* cam_emb,
                                 const int32_t* ray_cameras, int emb_dim, int use_emb, const void* w1, const void* w2,
                                 const void* w3, const float* d_rgb, const float* d_sigma, float density_scale, int64_t n,
                                 float* d_h, float* d_w1, float* d_w2, float* d_w3, float* d_cam_emb, int32_t* flags,
                                 int mlp_bf16, int geometry_feature_dim, int n_hidden_color, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_COLOR_DEPTH(n_hidden_color, w2 && d_w2);
    HRF_CHECK_ARG(ray_dirs && sample_ray && h && w1 && w3 && d_rgb && d_h && d_w1 && d_w3 && flags, "NULL argument");
    HRF_CHECK_COLOR_DIMS(geometry_feature_dim, emb_dim);
    const int G = geometry_feature_dim;
    HRF_CHECK_ARG(!(use_emb && emb_dim > 0) || (cam_emb && ray_cameras && d_cam_emb), "embedding requested without table");
    const int64_t tiles = (n + 15) / 16;
    unsigned blocks = (unsigned)((tiles + 3) / 4);
    const unsigned cap = n_hidden_color > 2 ? 256u : 512u;      // persistent: two workgroups per CU (one with three hidden layers)
    if (blocks > cap) blocks = cap;
    const int KT = (16 + G + emb_dim + 15) / 16;
#define HRF_LAUNCH_CB(K, PP, ET, NHC)                                                                                  \
    hipLaunchKernelGGL((k_mlp_bwd<K, PP, 2, NHC>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)nullptr,  \
                       ray_dirs, sample_ray, cam_emb, ray_cameras, emb_dim, (use_emb && emb_dim > 0) ? 1 : 0,             \
                       (const ET*)nullptr, (const ET*)nullptr, (const ET*)w1, (const ET*)w2, (const ET*)w3, density_scale, d_rgb, \
                       d_sigma, n, (void*)d_h, 1, (float*)nullptr, (float*)nullptr, d_w1, d_w2, d_w3,         \
                       d_cam_emb, flags, (const float*)nullptr, (const _Float16*)h, 0.0f, G)
#define HRF_LAUNCH_CB_P(NHC)                                                                                          \
    do {                                                                                                              \
        if (mlp_bf16) { if (KT == 2) HRF_LAUNCH_CB(2, Prec<true>, short, NHC); else HRF_LAUNCH_CB(3, Prec<true>, short, NHC); } \
        else { if (KT == 2) HRF_LAUNCH_CB(2, Prec<false>, _Float16, NHC); else HRF_LAUNCH_CB(3, Prec<false>, _Float16, NHC); }  \
    } while (0)
    HRF_DISPATCH_COLOR_DEPTH(n_hidden_color, HRF_LAUNCH_CB_P);
#undef HRF_LAUNCH_CB_P
#undef HRF_LAUNCH_CB
    HRF_CHECK_LAUNCH();
    return 0;
}
