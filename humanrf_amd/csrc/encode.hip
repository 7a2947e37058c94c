// Temporally-decomposed multi-resolution hash encoding for gfx950.
//
// Replaces Decomposition4D.forward/backward (humanrf/scene_representation/decomposition4d.py:124-135):
// four tcnn HashGrid encodings (xyz, xyt, yzt, xzt) and compose_tensors
// (humanrf/scene_representation/native/tensor_composition.cu:9-118) in ONE kernel, for samples of mixed
// temporal segments (per-sample segment id -> table base), so the four (N,32) half intermediates the
// reference materialises never exist unless the caller asks for them (training backward).
//
// Work mapping: workgroup = 256 threads = 4 wavefronts = one tile of 64 CONSECUTIVE samples (samples are
// sorted by ray and by distance along the ray, so the 64 lanes of a wavefront walk one or two rays and
// hit neighbouring cells: at coarse levels the 64 gathers of an instruction collapse to a few cache lines).
// Wavefront w owns levels {w, w+4, w+8, w+12} (interleaved so every wavefront gets the same mix of
// cheap coarse and expensive fine levels); a thread issues 4 levels x 4 encodings x 8 corners = 128
// independent 4-byte gathers. The composed features are staged through LDS and leave as 16-byte stores.
#include "encode_common.h"
#include "mlp_common.h"
#include <cstdlib>
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// query prep: positions, +0.5, frame -> (segment, local time)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_query_prep(
    const float* __restrict__ ray_origins, const float* __restrict__ ray_dirs, const int32_t* __restrict__ ray_frames,
    const int64_t* __restrict__ sample_ray, float* __restrict__ t_inout, const float* __restrict__ jitter, float step,
    const int32_t* __restrict__ frame_to_segment, const float* __restrict__ frame_to_local, int64_t n,
    float* __restrict__ out_xyzt, int32_t* __restrict__ out_segment)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = sample_ray[i];
    float t = t_inout[i];
    if (jitter) {
        t = t + jitter[i] * step;  // volume_rendering.py:63-64
        t_inout[i] = t;
    }
    const int32_t f = ray_frames[r];
    float4 q;
    // positions = o + t*d (volume_rendering.py:68-69), then +0.5 (humanrf.py:175)
    q.x = (ray_origins[r * 3 + 0] + t * ray_dirs[r * 3 + 0]) + 0.5f;
    q.y = (ray_origins[r * 3 + 1] + t * ray_dirs[r * 3 + 1]) + 0.5f;
    q.z = (ray_origins[r * 3 + 2] + t * ray_dirs[r * 3 + 2]) + 0.5f;
    q.w = frame_to_local[f];
    ((float4*)out_xyzt)[i] = q;
    out_segment[i] = frame_to_segment[f];
}

extern "C" int hrf_query_prep(const float* ray_origins, const float* ray_dirs, const int32_t* ray_frames,
                              const int64_t* sample_ray, float* t_inout, const float* jitter, float step,
                              const int32_t* frame_to_segment, const float* frame_to_local, int64_t n,
                              float* out_xyzt, int32_t* out_segment, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(ray_origins && ray_dirs && ray_frames && sample_ray && t_inout, "NULL input");
    HRF_CHECK_ARG(frame_to_segment && frame_to_local && out_xyzt && out_segment, "NULL table/output");
    hipLaunchKernelGGL(k_query_prep, dim3(hrf_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, ray_origins, ray_dirs,
                       ray_frames, sample_ray, t_inout, jitter, step, frame_to_segment, frame_to_local, n, out_xyzt,
                       out_segment);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// PD = void: the encoding alone. PD = Prec<bf16?>: sigma_net + truncated_exp ride along (hrf_encode4d_density_fwd, the render pass of
// the fused training step): the 64 samples' feature rows are in LDS when the gathers are done, each of the four wavefronts runs the
// 16 x 32 -> 64 -> 16 network on its quarter of them (6 MFMA, as the prune march does per step) and writes h and sigma -- values
// bit-identical to k_density_fwd on the stored features (same fragments, same instructions), without its launch and its re-read of
// the features (0.015 ms + a launch gap per step).
template <bool kSaveEnc, class PD = void>
__global__ __launch_bounds__(256, 4) void k_encode4d_fwd(   // 4 wavefronts per SIMD: 128 VGPRs
    
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const __half2* __restrict__ tables,
    const float* __restrict__ vectors, const hrf_segment_meta* __restrict__ segs, int vec_res, int64_t n,
    __half* __restrict__ out_features, __half* __restrict__ out_enc, int phase_shift,
    const void* __restrict__ dw1 = nullptr, const void* __restrict__ dw2 = nullptr, float density_scale = 0.0f,
    _Float16* __restrict__ out_h = nullptr, float* __restrict__ out_sigma = nullptr)
{
    constexpr bool kDensity = !std::is_void<PD>::value;
    typedef typename std::conditional<kDensity, PD, Prec<false>>::type PDx;
    __shared__ __attribute__((aligned(16))) typename PDx::E s_dw1[kDensity ? 64 * (32 + WPAD) : 1];
    __shared__ __attribute__((aligned(16))) typename PDx::E s_dw2[kDensity ? 16 * (64 + WPAD) : 1];
    if constexpr (kDensity) {      // (no barrier here: the one behind the level loop orders these stores before the reads)
        stage_rm(s_dw1, (const typename PDx::E*)dw1, 64, 32);
        stage_rm(s_dw2, (const typename PDx::E*)dw2, 16, 64);
    }
    __shared__ __attribute__((aligned(16))) __half2 tile[ENC_TILE][ENC_F / 2 + 4];  // +4: 16-B pad per row
    // per-encoding outputs (training only): [sample][encoding][level] half2, +4 pad per row of 64
    __shared__ __attribute__((aligned(16))) __half2 enc_tile[kSaveEnc ? ENC_TILE : 1][kSaveEnc ? 64 + 4 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long le_mask = (2ull << lane) - 1ull;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x takes the x-th eighth of the tiles. The training batch is
    // laid out by frame (hrf_pack_runs_sorted), so each 4 MB L2 sees the tables of one or two frames instead of all of them
    // (the schedule of the prune march, march.hip). The grid is a multiple of 8 workgroups.
    const int64_t tile_id = (int64_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t s = tile_id * ENC_TILE + lane;
    const bool valid = s < n;
    EncCoords q;
    int seg = 0;
    if (valid) {
        const float4 v = ((const float4*)xyzt)[s];
        q.c[0] = v.x; q.c[1] = v.y; q.c[2] = v.z; q.c[3] = v.w;
        seg = segment ? segment[s] : 0;
    } else {
        q.c[0] = q.c[1] = q.c[2] = q.c[3] = 0.0f;
    }
    int vc0[4], vc1[4];
    float vfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hrf_vec_tap(q.c[i], vec_res, vc0[i], vc1[i], vfr[i]);

    // The level loop, written once and instantiated twice: with the segment in a scalar register when the 64 samples of the
    // wavefront belong to ONE temporal segment (every wavefront but the few that straddle a segment boundary of the
    // frame-ordered training batch; always for one-segment models) -- level metadata are scalar loads, `hashed` is a scalar
    // branch, table addresses a scalar base + a 32-bit lane offset -- and with a per-lane segment otherwise. (Round 4 ran
    // every wavefront through the per-lane form: five vector loads of metadata per level, 64-bit lane addresses for all 32
    // gathers, and a DIVERGENT branch on `hashed` that dragged the dense levels' integer modulo through the hashed ones:
    // 8 100 lane-instructions per sample, profiles/r04_sq_k_encode4d_fwd.txt.)
    auto levels = [&](const hrf_segment_meta* sm, const float* vbase, int table_key) {
        const __half2* tbase = tables + sm->table_offset;
        const uint32_t entries = sm->entries;
        uint32_t li_done = 0;
#pragma unroll 1
        for (int lik = 0; lik < 4; ++lik) {
            const int li = enc_phase_next<4>(lik, phase_shift, li_done);   // (order: see enc_phase_next)
            const int l = wave + 4 * li;
            if (l >= (int)sm->n_levels) {
                // fewer than 16 levels: the row stays 32 wide -- ones up to the next multiple of 16 features (tcnn pads
                // sigma_net's input with ones), zeros beyond; the per-encoding outputs of a level that does not exist are zero
                const int ones_end = (2 * (int)sm->n_levels + 15) & ~15;
                const float cc = 2 * l < ones_end ? 1.0f : 0.0f;
                tile[lane][l] = __floats2half2_rn(cc, cc);
                if (kSaveEnc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) enc_tile[lane][e * 16 + l] = __floats2half2_rn(0.0f, 0.0f);
                }
                continue;
            }
            const hrf_level_meta lv = sm->levels[l];
            // lanes are consecutive samples of the ray-sorted batch: neighbours in the same cell share one fetch
            float feat[4][2];
#ifdef FWD_PLAIN_FROM_LEVEL
            if (l >= FWD_PLAIN_FROM_LEVEL) enc_level_plain(q, tbase, entries, lv, feat);   // (wave-uniform)
            else
#endif
            // (wide key on every level: adjacent lanes may be samples of DIFFERENT rays, and the packed 10-bit key of the march can alias
            // for coordinates outside [0,1] -- cell -1 next to cell 1023 of the neighbouring row)
            ENC_LEVEL_SHARED(q, tbase, entries, lv, le_mask, feat, table_key, true);
            if (kSaveEnc) {  // each tcnn encoding writes __half outputs (feat holds the rounded values)
#pragma unroll
                for (int e = 0; e < 4; ++e) enc_tile[lane][e * 16 + l] = __floats2half2_rn(feat[e][0], feat[e][1]);
            }
            // compose (tensor_composition.cu:47-54): xyz*v[3] + xyt*v[2] + yzt*v[0] + xzt*v[1]
            float sv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 v0 = *(const float2*)(vbase + (uint32_t)((i * vec_res + vc0[i]) * ENC_F + 2 * l));
                const float2 v1 = *(const float2*)(vbase + (uint32_t)((i * vec_res + vc1[i]) * ENC_F + 2 * l));
                sv[i][0] = v0.x + vfr[i] * (v1.x - v0.x);
                sv[i][1] = v0.y + vfr[i] * (v1.y - v0.y);
            }
            float r0 = ((feat[0][0] * sv[3][0] + feat[1][0] * sv[2][0]) + feat[2][0] * sv[0][0]) + feat[3][0] * sv[1][0];
            float r1 = ((feat[0][1] * sv[3][1] + feat[1][1] * sv[2][1]) + feat[2][1] * sv[0][1]) + feat[3][1] * sv[1][1];
            tile[lane][l] = __floats2half2_rn(r0, r1);
        }
    };
    const int seg0 = __builtin_amdgcn_readfirstlane(seg);
    if (__all(!valid || seg == seg0)) levels(segs + seg0, vectors + (size_t)seg0 * 4 * vec_res * ENC_F, 0);
    else levels(segs + seg, vectors + (size_t)seg * 4 * vec_res * ENC_F, seg + 1);
    __syncthreads();
    if constexpr (kDensity) {
        typedef typename PDx::V V;
        const int g = lane >> 4, c = lane & 15;
        const _Float16* feat = (const _Float16*)&tile[0][0];
        constexpr int ROWH = 2 * (ENC_F / 2 + 4);                         // halves per feature row of the tile
        const int64_t so = tile_id * ENC_TILE + 16 * wave + c;             // this lane's sample of the wavefront's 16
        const V xv0 = pv_from_h4<PDx>(*(const h4*)(feat + (16 * wave + c) * ROWH + 4 * g));
        const V xv1 = pv_from_h4<PDx>(*(const h4*)(feat + (16 * wave + c) * ROWH + 16 + 4 * g));
        V hid[4];
#pragma unroll
        for (int ht = 0; ht < 4; ++ht)
            hid[ht] = pv_relu<PDx>(PDx::mfma2(afrag(s_dw1, 32, ht, 0, lane), afrag(s_dw1, 32, ht, 1, lane), xv0, xv1, f4zero()));
        f4 o = PDx::mfma2(afrag(s_dw2, 64, 0, 0, lane), afrag(s_dw2, 64, 0, 1, lane), hid[0], hid[1], f4zero());
        o = PDx::mfma2(afrag(s_dw2, 64, 0, 2, lane), afrag(s_dw2, 64, 0, 3, lane), hid[2], hid[3], o);
        if (so < n) {
            f4 orr;   // the network's output rounded to its 16-bit type, then carried in an fp16 container (as k_density_fwd)
#pragma unroll
            for (int r = 0; r < 4; ++r) orr[r] = p_round<PDx>(o[r]);
            const h4 oh = to_h4(orr);
            if (out_h) *(h4*)(out_h + so * 16 + 4 * g) = oh;
            if (out_sigma && g == 0) out_sigma[so] = expf((float)oh[0]) * density_scale;
        }
    }
    // 64 samples x 64 B -> 256 threads x 16 B, fully coalesced
    {
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int64_t so = tile_id * ENC_TILE + row;
        if (so < n) {
            const uint4 v = *(const uint4*)&tile[row][part * 4];
            *(uint4*)(out_features + so * ENC_F + part * 8) = v;
        }
    }
    if (kSaveEnc) {
        // 64 samples x 256 B: each thread moves 4 x 16 B, a wavefront writes 1 KiB contiguous per step
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int chunk = it * 256 + threadIdx.x;      // 16-byte chunk index within the tile (1024 chunks)
            const int row = chunk >> 4, part = chunk & 15;
            const int64_t so = tile_id * ENC_TILE + row;
            if (so < n) {
                const uint4 v = *(const uint4*)&enc_tile[row][part * 4];
                *(uint4*)(out_enc + so * 4 * ENC_F + part * 8) = v;
            }
        }
    }
}

extern "C" int hrf_encode4d_fwd(const float* xyzt, const int32_t* segment, const void* tables, const float* vectors,
                                const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                void* out_features, void* out_enc_features, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(xyzt && tables && vectors && segments && out_features, "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1, "bad segment count / vector resolution");
    dim3 grid((hrf_blocks(n, ENC_TILE) + 7u) & ~7u), block(256);   // whole rounds over the 8 XCDs
    int phase_shift = ENC_PHASE_SHIFT;   // (see enc_phase_next, encode_common.h)
#ifdef ENC_PHASE_TUNE
    if (const char* e = getenv("HRF_PHASE_SHIFT_FWD")) phase_shift = atoi(e);
#endif
    if (out_enc_features)
        hipLaunchKernelGGL(k_encode4d_fwd<true>, grid, block, 0, (hipStream_t)stream, xyzt, segment,
                           (const __half2*)tables, vectors, segments, vec_res, n, (__half*)out_features,
                           (__half*)out_enc_features, phase_shift);
    else
        hipLaunchKernelGGL(k_encode4d_fwd<false>, grid, block, 0, (hipStream_t)stream, xyzt, segment,
                           (const __half2*)tables, vectors, segments, vec_res, n, (__half*)out_features,
                           (__half*)nullptr, phase_shift);
    HRF_CHECK_LAUNCH();
    return 0;
}

// hrf_encode4d_fwd + hrf_density_mlp_fwd in one launch (ABI 8): the render pass of the fused training step.
extern "C" int hrf_encode4d_density_fwd(const float* xyzt, const int32_t* segment, const void* tables, const float* vectors,
                                        const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                        void* out_features, void* out_enc_features, const void* w1, const void* w2,
                                        float density_scale, void* out_h, float* out_sigma, int mlp_bf16, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(xyzt && tables && vectors && segments && out_features && out_enc_features, "NULL argument");
    HRF_CHECK_ARG(w1 && w2 && (out_h || out_sigma), "NULL network argument");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1, "bad segment count / vector resolution");
    dim3 grid((hrf_blocks(n, ENC_TILE) + 7u) & ~7u), block(256);   // whole rounds over the 8 XCDs
    int phase_shift = ENC_PHASE_SHIFT;
#ifdef ENC_PHASE_TUNE
    if (const char* e = getenv("HRF_PHASE_SHIFT_FWD")) phase_shift = atoi(e);
#endif
    if (mlp_bf16)
        hipLaunchKernelGGL((k_encode4d_fwd<true, Prec<true>>), grid, block, 0, (hipStream_t)stream, xyzt, segment, (const __half2*)tables,
                           vectors, segments, vec_res, n, (__half*)out_features, (__half*)out_enc_features, phase_shift, w1, w2,
                           density_scale, (_Float16*)out_h, out_sigma);
    else
        hipLaunchKernelGGL((k_encode4d_fwd<true, Prec<false>>), grid, block, 0, (hipStream_t)stream, xyzt, segment, (const __half2*)tables,
                           vectors, segments, vec_res, n, (__half*)out_features, (__half*)out_enc_features, phase_shift, w1, w2,
                           density_scale, (_Float16*)out_h, out_sigma);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward: table gradients (tcnn kernel_grid_backward semantics, fp32 accumulation instead of __half2
// atomics) and 1-D vector gradients (tensor_composition.cu:85-117).
//
// A naive scatter (one atomic per sample x corner x feature) serialises in the L2 atomic units: samples are
// consecutive along a ray, so at coarse levels the 64 lanes of an instruction hit the same few addresses.
// Both kernels therefore aggregate ALONG THE RAY before touching memory: a thread walks BWD_RUN consecutive
// samples, keeps the gradient of the current cell (8 corners x 2 features) / vector tap in registers and only
// issues atomics when the cell / tap changes. Level is wavefront-uniform, so all lanes of a wavefront flush at
// a similar cadence (coarse levels: once per run; finest level: nearly every sample).
// ------------------------------------------------------------------------------------------------
#define BWD_TILE 256   // samples per workgroup
#define BWD_RUN 16     // consecutive samples walked by one thread
// dwords per LDS row of dY (+1 pad: conflict-free column reads); kF32: dY arrives as fp32 (fused training path)
// instead of the __half tensor autograd hands to a stand-alone Decomposition4D.
template <bool kF32>
__global__ __launch_bounds__(256) void k_encode4d_bwd_tables(
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const float* __restrict__ vectors,
    const hrf_segment_meta* __restrict__ segs, int vec_res, int64_t n, const void* __restrict__ d_features,
    float inv_scale, float* __restrict__ d_tables, float gb, int32_t* __restrict__ flags)
{
    const float inv_gb = gb > 0.0f ? 1.0f / gb : 0.0f;
    constexpr int ROW = kF32 ? 32 : 16;
    constexpr int BWD_DY_STRIDE = ROW + 1;
    __shared__ float4 s_q[BWD_TILE];
    __shared__ int s_seg[BWD_TILE];
    __shared__ uint32_t s_dy[BWD_TILE * BWD_DY_STRIDE];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * BWD_TILE;
    const int n_here = (int)min((int64_t)BWD_TILE, n - base);
    if (tid < n_here) {
        s_q[tid] = ((const float4*)xyzt)[base + tid];
        s_seg[tid] = segment ? segment[base + tid] : 0;
    }
    for (int i = tid; i < n_here * ROW; i += 256) {
        const int row = i / ROW, col = i % ROW;
        s_dy[row * BWD_DY_STRIDE + col] = ((const uint32_t*)d_features)[(base + row) * ROW + col];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int chunk = lane & 15, e = lane >> 4;      // 16 chunks of BWD_RUN samples x 4 encodings per wavefront
    const int vi = (e == 0) ? 3 : (e == 1) ? 2 : (e == 2) ? 0 : 1;  // vector paired with encoding e
    const int s0 = chunk * BWD_RUN;
    if (s0 >= n_here) return;
    const int s1 = min(s0 + BWD_RUN, n_here);
#pragma unroll 1
    for (int li = 0; li < 4; ++li) {
        const int l = wave + 4 * li;
        // The gradient of this encoding's output, d_feat_e = v[pair(e)] * dY (tensor_composition.cu:112-115), is kept
        // in fp32 (heavily shared coarse entries sum thousands of these terms). Its vector taps are global loads
        // that do not depend on the cell walk: they are fetched one sample ahead so their latency overlaps the
        // walk of the current sample.
        float acc[8][2];
        uint32_t cidx[8];
        uint32_t pa = 0xFFFFFFFFu, pb = 0, pc = 0;
        int pseg = -1;
        float* tg = nullptr;
        bool have = false;
        hrf_level_meta lv;
        lv.scale = 0; lv.res = 1; lv.size = 1; lv.offset = 0; lv.hashed = 0;
        float2 nv0, nv1;
        float nfr;
        {
            const float4 q4 = s_q[s0];
            const float cvi = (vi == 0) ? q4.x : (vi == 1) ? q4.y : (vi == 2) ? q4.z : q4.w;
            int c0, c1;
            hrf_vec_tap(cvi, vec_res, c0, c1, nfr);
            const float* vb = vectors + ((size_t)s_seg[s0] * 4 + vi) * vec_res * ENC_F + 2 * l;
            nv0 = *(const float2*)(vb + (size_t)c0 * ENC_F);
            nv1 = *(const float2*)(vb + (size_t)c1 * ENC_F);
        }
#pragma unroll 1
        for (int s = s0; s < s1; ++s) {
            const float4 q4 = s_q[s];
            const int seg = s_seg[s];
            const float2 v0 = nv0, v1 = nv1;
            const float fr = nfr;
            if (s + 1 < s1) {
                const float4 qn = s_q[s + 1];
                const float cvi = (vi == 0) ? qn.x : (vi == 1) ? qn.y : (vi == 2) ? qn.z : qn.w;
                int c0, c1;
                hrf_vec_tap(cvi, vec_res, c0, c1, nfr);
                const float* vb = vectors + ((size_t)s_seg[s + 1] * 4 + vi) * vec_res * ENC_F + 2 * l;
                nv0 = *(const float2*)(vb + (size_t)c0 * ENC_F);
                nv1 = *(const float2*)(vb + (size_t)c1 * ENC_F);
            }
            float2 dy;
            if (kF32) {
                dy.x = __uint_as_float(s_dy[s * BWD_DY_STRIDE + 2 * l]);
                dy.y = __uint_as_float(s_dy[s * BWD_DY_STRIDE + 2 * l + 1]);
            } else {
                const uint32_t dyu = s_dy[s * BWD_DY_STRIDE + l];
                dy = __half22float2(*(const __half2*)&dyu);
            }
            const float g0 = hrf_through_half((v0.x + fr * (v1.x - v0.x)) * dy.x * inv_scale, gb, inv_gb);
            const float g1 = hrf_through_half((v0.y + fr * (v1.y - v0.y)) * dy.y * inv_scale, gb, inv_gb);
            // a value the half boundary turns into inf (|x / gb| > 65504), or one that arrives non-finite: the step must be
            // skipped like GradScaler skips it (the binned scatter catches the same through its range check)
            if (flags && (!(fabsf(g0) < 3.0e38f) || !(fabsf(g1) < 3.0e38f))) atomicOr(flags, 1);
            if (seg != pseg) {
                if (l >= (int)segs[seg].n_levels) continue;
                lv = segs[seg].levels[l];
            }
            EncCoords q; q.c[0] = q4.x; q.c[1] = q4.y; q.c[2] = q4.z; q.c[3] = q4.w;
            float a, b, c;
            enc_pick(q, e, a, b, c);
            const float fpa = fmaf(a, lv.scale, 0.5f), fpb = fmaf(b, lv.scale, 0.5f), fpc = fmaf(c, lv.scale, 0.5f);
            const float fa = floorf(fpa), fb = floorf(fpb), fc = floorf(fpc);
            const uint32_t ia = (uint32_t)(int)fa, ib = (uint32_t)(int)fb, ic = (uint32_t)(int)fc;
            if (!have || ia != pa || ib != pb || ic != pc || seg != pseg) {
                if (have) {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        if (acc[kk][0] != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx[kk], acc[kk][0]);
                        if (acc[kk][1] != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx[kk] + 1, acc[kk][1]);
                    }
                }
                const hrf_segment_meta* sm = segs + seg;
                tg = d_tables + 2 * (sm->table_offset + (size_t)e * sm->entries + lv.offset);
                Corner8 cr;
                enc_corners(a, b, c, lv, cr);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { cidx[kk] = cr.idx[kk]; acc[kk][0] = 0.0f; acc[kk][1] = 0.0f; }
                pa = ia; pb = ib; pc = ic; pseg = seg; have = true;
            }
            const float wa = fpa - fa, wb = fpb - fb, wc = fpc - fc;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float w = 1.0f;
                w *= (kk & 1) ? wa : (1.0f - wa);
                w *= (kk & 2) ? wb : (1.0f - wb);
                w *= (kk & 4) ? wc : (1.0f - wc);
                acc[kk][0] = fmaf(w, g0, acc[kk][0]);
                acc[kk][1] = fmaf(w, g1, acc[kk][1]);
            }
        }
        if (have) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (acc[kk][0] != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx[kk], acc[kk][0]);
                if (acc[kk][1] != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx[kk] + 1, acc[kk][1]);
            }
        }
    }
}

// Level-major variant used by the fused training path.
//
// Measured on MI355X (tools/microbench/atomic_bench.hip): non-returning global atomics are limited by REQUESTS, about
// 21 G/s chip-wide whatever the footprint (2 MB or 256 MB) and whatever the type (fp32 or packed half2); lanes of
// one instruction that hit contiguous dwords are merged into one request (2/4/16 adjacent lanes: 42/84/330 G
// lane-atomics/s), and lanes hitting one address serialise (0.5 G/s). So the scatter is organised to issue as
// few, as wide requests as possible:
//   * 16 lanes cooperate on one cell: lane j owns (corner j>>1, feature j&1), so the two features of an entry and
//     -- whenever the x-neighbour entry is adjacent (always on dense levels, for even x on hashed ones) -- both
//     x-corners leave in ONE instruction as 16 contiguous bytes: 4..8 requests per cell instead of 16;
//   * gradients are accumulated ALONG THE RAY in registers and flushed only when the cell changes (samples are
//     consecutive along rays), which also removes the same-address serialisation at coarse levels;
//   * dY arrives as fp32 in level-major layout dY_lm[level][sample] = (dY[2l], dY[2l+1]) (written that way by
//     k_mlp_bwd) and the grid is ordered level-major, so at any moment the chip scatters into the tables of one or
//     two levels only.
// Wavefront = one run of LM_RUN consecutive samples x 4 encodings x 16 (corner, feature) lanes.
#define LM_TILE 256
#ifndef LM_RUN
#define LM_RUN 16  // consecutive samples walked by one wavefront (32 / 64 measured: no change, 1.41-1.48 ms)
#endif
// One LDS record per (sample, encoding): everything a lane of that encoding needs for one step of the walk, so that the
// walk issues two 16-byte LDS reads from ONE running address instead of five reads from five.
struct LmRec {
    uint32_t ia, ib, ic;   // cell coordinates of the encoding's three axes (tcnn pos_fract: floor(fma(c, scale, 0.5)))
    float g0;              // upstream gradient of the encoding's output, feature 0: v[pair(e)][0] * dY[0] / grad_scale
    float wa, wb, wc;      // fractions
    float g1;              // ... feature 1
};
template <int LM_TILE_T>
__global__ __launch_bounds__(256) void k_encode4d_bwd_tables_lm(
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const float* __restrict__ vectors,
    const hrf_segment_meta* __restrict__ segs, int vec_res, int64_t n, const float* __restrict__ dY_lm,
    float inv_scale, float* __restrict__ d_tables, int64_t n_tiles, float gb, int32_t* __restrict__ flags)
{
    const float inv_gb = gb > 0.0f ? 1.0f / gb : 0.0f;
    // The sequential walk along the samples is bound by vector-ALU issue (one sample per wavefront iteration), so
    // everything a sample contributes that does not depend on the walk is computed ONCE per workgroup, with one
    // thread per sample, and parked in LDS: per axis the cell coordinate and the fraction of this level, and per
    // (encoding, feature) the upstream gradient of the encoding's output, d_feat_e[f] = v[pair(e)][f] * dY[f]
    // (tensor_composition.cu:112-115, kept in fp32) -- laid out as one 32-byte record per (sample, encoding).
    // Measured (678 k samples of 43 k rays, 16 per ray; profiles/r02_microbench_scatter_probe.txt): this form,
    // the previous one with five LDS arrays and ~2x the instructions per step, and 128-sample tiles all take 2.1-2.2 ms; the
    // same walk WITHOUT its atomic instruction takes 0.92 ms. The kernel is bound by L2 atomic requests (PMC: 58 per sample
    // at 16 samples per ray = 0.86 of the 21 G requests/s the chip sustains), and their number follows the rays per batch:
    // every ray opens new cells on every level.
    __shared__ int s_seg[LM_TILE_T];
    __shared__ __attribute__((aligned(16))) LmRec s_rec[4][LM_TILE_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = (int)(blockIdx.x / n_tiles);
    const int64_t base = (blockIdx.x % n_tiles) * LM_TILE_T;
    const int n_here = (int)min((int64_t)LM_TILE_T, n - base);
    if (tid < n_here) {
        const float4 q4 = ((const float4*)xyzt)[base + tid];
        const int sg = segment ? segment[base + tid] : 0;
        const bool has_level = l < (int)segs[sg].n_levels;
        s_seg[tid] = has_level ? sg : -1;
        if (has_level) {
            const float scale = segs[sg].levels[l].scale;
            const float qc[4] = {q4.x, q4.y, q4.z, q4.w};
            const float2 dy = *(const float2*)(dY_lm + ((size_t)l * n + base + tid) * 2);
            float sv[4][2], w[4];
            uint32_t ci[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float p = fmaf(qc[v], scale, 0.5f);
                const float fl = floorf(p);
                ci[v] = (uint32_t)(int)fl;
                w[v] = p - fl;
                int c0, c1;
                float fr;
                hrf_vec_tap(qc[v], vec_res, c0, c1, fr);
                const float* vb = vectors + ((size_t)(sg * 4 + v) * vec_res) * ENC_F + 2 * l;
                const float2 v0 = *(const float2*)(vb + (size_t)c0 * ENC_F), v1 = *(const float2*)(vb + (size_t)c1 * ENC_F);
                sv[v][0] = v0.x + fr * (v1.x - v0.x);
                sv[v][1] = v0.y + fr * (v1.y - v0.y);
            }
            // encoding e: axes (a,b,c) = xyz, xyt, yzt, xzt; pairs with vector {3, 2, 0, 1}[e] (tensor_composition.cu:47-54)
            const int ax[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
            const int pv[4] = {3, 2, 0, 1};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                LmRec r;
                r.ia = ci[ax[e][0]]; r.ib = ci[ax[e][1]]; r.ic = ci[ax[e][2]];
                r.wa = w[ax[e][0]]; r.wb = w[ax[e][1]]; r.wc = w[ax[e][2]];
                r.g0 = hrf_through_half(sv[pv[e]][0] * dy.x * inv_scale, gb, inv_gb);
                r.g1 = hrf_through_half(sv[pv[e]][1] * dy.y * inv_scale, gb, inv_gb);
                if (flags && (!(fabsf(r.g0) < 3.0e38f) || !(fabsf(r.g1) < 3.0e38f))) atomicOr(flags, 1);   // (see k_encode4d_bwd_tables)
                s_rec[e][tid] = r;
            }
        }
    }
    __syncthreads();
    const int e = lane >> 4, j = lane & 15;
    const int f = j & 1, cx = (j >> 1) & 1, cy = (j >> 2) & 1, cz = (j >> 3) & 1;
    const LmRec* rec = s_rec[e];
    // lane-constant forms of the corner weights: (c ? w : 1 - w) == fma(w, s, b) with (s, b) = (1, 0) or (-1, 1), exactly
    const float sxw = cx ? 1.0f : -1.0f, bxw = cx ? 0.0f : 1.0f;
    const float syw = cy ? 1.0f : -1.0f, byw = cy ? 0.0f : 1.0f;
    const float szw = cz ? 1.0f : -1.0f, bzw = cz ? 0.0f : 1.0f;
    // a role (cx,cy,cz) moves to the neighbouring lane of its group when the cell moves by one along an axis
    const int sgn_x = 2 * cx - 1, sgn_y = 2 * cy - 1, sgn_z = 2 * cz - 1;

    // Run length by level: every run ends with a flush of its last cell (8 corners), which is pure overhead at coarse levels
    // where 64 march samples stay inside one or two cells (level 0: 0.012 cells per sample) and nothing at the finest
    // ones where every sample is a new cell anyway.
    const int run_len = min((l < 6) ? 4 * LM_RUN : (l < 12) ? 2 * LM_RUN : LM_RUN, LM_TILE_T / 4);
#pragma unroll 1
    for (int run = wave; run * run_len < n_here; run += 4) {
        const int s0 = run * run_len, s1 = min(s0 + run_len, n_here);
        float acc = 0.0f;
        uint32_t cidx = 0;
        uint32_t pa = 0, pb = 0, pc = 0;
        int pseg = -1, seg_loaded = -2;
        float* tg = nullptr;
        float* tg_seg = nullptr;
        bool have = false;
        uint32_t lv_res = 1, lv_size = 1, lv_hashed = 0;
#pragma unroll 1
        for (int s = s0; s < s1; ++s) {
            const int seg = s_seg[s];
            const uint4 r0 = *(const uint4*)&rec[s];                 // ia, ib, ic, g0
            const float4 r1 = *((const float4*)&rec[s] + 1);         // wa, wb, wc, g1
            if (seg != seg_loaded) {  // segment metadata: fetched when the segment changes, not at every cell change
                if (seg < 0) continue;   // this sample's segment has fewer levels
                const hrf_level_meta lv = segs[seg].levels[l];
                lv_res = lv.res; lv_size = lv.size; lv_hashed = lv.hashed;
                tg_seg = d_tables + 2 * (segs[seg].table_offset + (size_t)e * segs[seg].entries + lv.offset);
                seg_loaded = seg;
            }
            const uint32_t ia = r0.x, ib = r0.y, ic = r0.z;
            const float gval = f ? r1.w : __uint_as_float(r0.w);
            if (!have || ia != pa || ib != pb || ic != pc || seg != pseg) {
                // Cell change. Neighbouring cells share corners: when the walk moves by at most one cell per axis
                // (the usual case at fine levels: the march step is ~0.8 of the finest cell), the corner this lane
                // now owns may be a corner another lane of the group was already accumulating -- take that lane's
                // running sum instead of flushing it, and flush only corners the new cell no longer touches.
                const int mx = (int)(ia - pa), my = (int)(ib - pb), mz = (int)(ic - pc);
                const bool adjacent = have && seg == pseg && (unsigned)(mx + 1) <= 2u && (unsigned)(my + 1) <= 2u &&
                                      (unsigned)(mz + 1) <= 2u;
                // old role c survives iff c - m is a corner of the new cell: m == 0 or m == 2c-1
                const bool kept_by_new = adjacent && (mx == 0 || mx == sgn_x) && (my == 0 || my == sgn_y) && (mz == 0 || mz == sgn_z);
                if (have && !kept_by_new && acc != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx + f, acc);
                // new role c continues old role c + m when that is a corner of the old cell: m == 0 or m == 1-2c;
                // that role sits in the lane whose corner bit is flipped on every axis that moved
                const bool inherits = adjacent && (mx == 0 || mx == -sgn_x) && (my == 0 || my == -sgn_y) && (mz == 0 || mz == -sgn_z);
                const int flip = ((mx != 0) ? 2 : 0) | ((my != 0) ? 4 : 0) | ((mz != 0) ? 8 : 0);
                const float carried = __shfl(acc, inherits ? (lane ^ flip) : lane, 64);
                tg = tg_seg;
                {   // same index as enc_corners (tcnn grid_index): mask on hashed levels (size is a power of two there),
                    // on dense levels the stride form, which only wraps for the far corner of the last cell -- no
                    // integer division on the common path (the generic `% size` costs ~16 VALU instructions)
                    const uint32_t x = ia + cx, y = ib + cy, z = ic + cz;
                    if (lv_hashed) {
                        cidx = (x ^ (y * 2654435761u) ^ (z * 805459861u)) & (lv_size - 1u);
                    } else {
                        uint32_t i = x + y * lv_res + z * lv_res * lv_res;
                        if (i >= lv_size) { i -= lv_size; if (i >= lv_size) i %= lv_size; }
                        cidx = i;
                    }
                }
                acc = inherits ? carried : 0.0f;
                pa = ia; pb = ib; pc = ic; pseg = seg; have = true;
            }
            // w = (cx ? wa : 1-wa) * (cy ? wb : 1-wb) * (cz ? wc : 1-wc), same operations and order as before
            float w = 1.0f * fmaf(r1.x, sxw, bxw);
            w *= fmaf(r1.y, syw, byw);
            w *= fmaf(r1.z, szw, bzw);
            acc = fmaf(w, gval, acc);
        }
        if (have && acc != 0.0f) unsafeAtomicAdd(tg + 2 * (size_t)cidx + f, acc);
    }
}

// d_vectors[vi][c0|c1][f] += feat_pair(vi)[f] * dY[f] * (1-fr | fr)  (tensor_composition.cu:97-108).
// Thread = (run of VEC_RUN consecutive samples, feature f): the 32 features of a tap row are 128 contiguous
// bytes, so a half-wavefront's atomics land in one line; a run keeps the two taps of every vector in registers
// until the tap index moves. Run length measured on MI355X (640 k samples, requests bound it): 8 -> 0.41 ms,
// 16 -> 0.29, 32 -> 0.25, 64 -> 0.21, 128 -> 0.28 (too few wavefronts).
#define VEC_TILE 512
#define VEC_RUN 64

// kMode: 0 = __half [n][32], 1 = fp32 [n][32], 2 = fp32 level-major [16][n][2]
template <int kMode>
__global__ __launch_bounds__(256) void k_encode4d_bwd_vectors(
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const __half* __restrict__ enc_feats,
    int vec_res, int64_t n, const void* __restrict__ d_features, float inv_scale, float* __restrict__ d_vectors)
{
    __shared__ float s_flush[512];
    __shared__ int s_key[3];
    const int f = threadIdx.x & 31, run = threadIdx.x >> 5;  // 8 runs x 32 features
    const int64_t s0 = min((int64_t)blockIdx.x * VEC_TILE + (int64_t)run * VEC_RUN, n);   // (an empty run still meets the barriers)
    const int64_t s1 = min(s0 + (int64_t)VEC_RUN, n);
    // feature order of encodings in enc_feats: xyz, xyt, yzt, xzt; vector vi pairs with {yzt, xzt, xyt, xyz}
    const int enc_of_vi[4] = {2, 3, 1, 0};
    float acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
    int pc0[4] = {-1, -1, -1, -1}, pc1[4] = {-1, -1, -1, -1};
    int pseg = -1;
    // one sample of look-ahead: the loads of sample s+1 are in flight while sample s is accumulated / flushed
    struct In { float4 q; int seg; float dy; __half e[4]; };
    auto fetch = [&](int64_t s) {
        In v;
        v.q = ((const float4*)xyzt)[s];
        v.seg = segment ? segment[s] : 0;
        if (kMode == 0) v.dy = __half2float(((const __half*)d_features)[s * ENC_F + f]);
        else if (kMode == 1) v.dy = ((const float*)d_features)[s * ENC_F + f];
        else v.dy = ((const float*)d_features)[((size_t)(f >> 1) * n + s) * 2 + (f & 1)];
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) v.e[vi] = enc_feats[(s * 4 + enc_of_vi[vi]) * ENC_F + f];
        return v;
    };
    In nxt;
    if (s0 < s1) nxt = fetch(s0);
    for (int64_t s = s0; s < s1; ++s) {
        const In cur = nxt;
        if (s + 1 < s1) nxt = fetch(s + 1);
        const float qc[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w};
        const int seg = cur.seg;
        const float dy = cur.dy * inv_scale;
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            int c0, c1; float fr;
            hrf_vec_tap(qc[vi], vec_res, c0, c1, fr);
            if (c0 != pc0[vi] || c1 != pc1[vi] || seg != pseg) {
                // samples walk along a ray: the taps usually move by ONE row, and the row both samples share keeps
                // its partial sum in registers (one flush per row crossed instead of two per tap change)
                const bool same_vec = seg == pseg && pc0[vi] >= 0;
                const bool fwd1 = same_vec && c0 == pc1[vi] && pc0[vi] != pc1[vi];
                const bool bwd1 = same_vec && c1 == pc0[vi] && pc0[vi] != pc1[vi];
                if (pc0[vi] >= 0) {
                    float* row = d_vectors + ((size_t)pseg * 4 + vi) * vec_res * ENC_F + f;
                    if (!bwd1 && acc0[vi] != 0.0f) unsafeAtomicAdd(row + (size_t)pc0[vi] * ENC_F, acc0[vi]);
                    if (!fwd1 && acc1[vi] != 0.0f) unsafeAtomicAdd(row + (size_t)pc1[vi] * ENC_F, acc1[vi]);
                }
                const float keep0 = fwd1 ? acc1[vi] : 0.0f, keep1 = bwd1 ? acc0[vi] : 0.0f;
                pc0[vi] = c0; pc1[vi] = c1; acc0[vi] = keep0; acc1[vi] = keep1;
            }
            const float dval = __half2float(cur.e[vi]) * dy;
            acc0[vi] = fmaf(dval, 1.0f - fr, acc0[vi]);
            acc1[vi] = fmaf(dval, fr, acc1[vi]);
        }
        pseg = seg;  // updated after all four vectors compared against the previous sample's segment
    }
#pragma unroll
    for (int vi = 0; vi < 3; ++vi) {
        if (pc0[vi] >= 0) {
            float* row = d_vectors + ((size_t)pseg * 4 + vi) * vec_res * ENC_F + f;
            if (acc0[vi] != 0.0f) unsafeAtomicAdd(row + (size_t)pc0[vi] * ENC_F, acc0[vi]);
            if (acc1[vi] != 0.0f) unsafeAtomicAdd(row + (size_t)pc1[vi] * ENC_F, acc1[vi]);
        }
    }
    // The TIME vector (vi = 3): every sample of a frame taps the same two rows, and a batch laid out by frame sends whole
    // workgroups -- thousands of them in a row -- to those 64 addresses (one hot address retires 0.5 G atomics/s,
    // profiles/r01_microbench_atomic_rates.txt: measured 0.30 ms for this kernel on the sorted batch against 0.21 in draw
    // order). The eight runs of a workgroup that end on the rows of its first run are summed in LDS and leave as ONE atomic
    // per (row, feature); the others (a frame boundary inside the tile) go directly.
    s_flush[threadIdx.x] = 0.0f;
    s_flush[256 + threadIdx.x] = 0.0f;
    if (run == 0 && f == 0) { s_key[0] = pseg; s_key[1] = pc0[3]; s_key[2] = pc1[3]; }
    __syncthreads();
    const bool live = pc0[3] >= 0;
    const bool shared = live && pseg == s_key[0] && pc0[3] == s_key[1] && pc1[3] == s_key[2];
    if (shared) {
        s_flush[run * 32 + f] = acc0[3];
        s_flush[256 + run * 32 + f] = acc1[3];
    } else if (live) {
        float* row = d_vectors + ((size_t)pseg * 4 + 3) * vec_res * ENC_F + f;
        if (acc0[3] != 0.0f) unsafeAtomicAdd(row + (size_t)pc0[3] * ENC_F, acc0[3]);
        if (acc1[3] != 0.0f) unsafeAtomicAdd(row + (size_t)pc1[3] * ENC_F, acc1[3]);
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_key[1] >= 0) {
        const int tap = threadIdx.x >> 5, ff = threadIdx.x & 31;
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += s_flush[tap * 256 + r * 32 + ff];
        if (sum != 0.0f)
            unsafeAtomicAdd(d_vectors + (((size_t)s_key[0] * 4 + 3) * vec_res + (size_t)(tap ? s_key[2] : s_key[1])) * ENC_F + ff, sum);
    }
}

extern "C" int hrf_encode4d_bwd(const float* xyzt, const int32_t* segment, const void* enc_features,
                                const float* vectors, const hrf_segment_meta* segments, int num_segments, int vec_res,
                                int64_t n, const void* d_features, int d_features_mode, float grad_scale,
                                float grad_boundary, float* d_tables, float* d_vectors, int32_t* flags, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(grad_boundary >= 0.0f, "grad_boundary must be 0 (off) or the factor between the fused and the reference's gradient scale");
    const float gb = grad_boundary;
    HRF_CHECK_ARG(xyzt && enc_features && vectors && segments && d_features && (d_tables || d_vectors), "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1 && grad_scale > 0.0f, "bad arguments");
    HRF_CHECK_ARG(d_features_mode >= 0 && d_features_mode <= 2, "d_features_mode must be 0 (fp16), 1 (fp32) or 2 (fp32 level-major)");
    const dim3 gt(hrf_blocks(n, BWD_TILE)), gv(hrf_blocks(n, VEC_TILE)), blk(256);
    const float inv = 1.0f / grad_scale;
    hipStream_t st = (hipStream_t)stream;
    if (!d_tables) {
    } else if (d_features_mode == 2) {
        // (samples per workgroup 128 instead of 256 and per-level launches were measured in round 2 with build-time probes:
        // profiles/r02_microbench_scatter_probe.txt; the library carries no measurement switches)
        const int64_t n_tiles = (n + 255) / 256;
        hipLaunchKernelGGL(k_encode4d_bwd_tables_lm<256>, dim3((unsigned)(n_tiles * HRF_MAX_LEVELS)), blk, 0, st, xyzt, segment,
                           vectors, segments, vec_res, n, (const float*)d_features, inv, d_tables, n_tiles, gb, flags);
    } else if (d_features_mode == 1) {
        hipLaunchKernelGGL(k_encode4d_bwd_tables<true>, gt, blk, 0, st, xyzt, segment, vectors, segments, vec_res, n,
                           d_features, inv, d_tables, gb, flags);
    } else {
        hipLaunchKernelGGL(k_encode4d_bwd_tables<false>, gt, blk, 0, st, xyzt, segment, vectors, segments, vec_res, n,
                           d_features, inv, d_tables, gb, flags);
    }
    HRF_CHECK_LAUNCH();
    if (!d_vectors) return 0;
    if (d_features_mode == 2)
        hipLaunchKernelGGL(k_encode4d_bwd_vectors<2>, gv, blk, 0, st, xyzt, segment, (const __half*)enc_features,
                           vec_res, n, d_features, inv, d_vectors);
    else if (d_features_mode == 1)
        hipLaunchKernelGGL(k_encode4d_bwd_vectors<1>, gv, blk, 0, st, xyzt, segment, (const __half*)enc_features,
                           vec_res, n, d_features, inv, d_vectors);
    else
        hipLaunchKernelGGL(k_encode4d_bwd_vectors<0>, gv, blk, 0, st, xyzt, segment, (const __half*)enc_features,
                           vec_res, n, d_features, inv, d_vectors);
    HRF_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone compose op with the reference's own signature (tensor_composition.cu:120-225)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compose_fwd(
    const __half* __restrict__ f_xyz, const __half* __restrict__ f_xyt, const __half* __restrict__ f_yzt,
    const __half* __restrict__ f_xzt, const float* __restrict__ vectors, const float* __restrict__ xyzt, int64_t n,
    int F, int Rv, __half* __restrict__ out)
{
    const int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= n * F) return;
    const int f = (int)(index % F);
    const int64_t s = index / F;
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c0, c1; float fr;
        hrf_vec_tap(xyzt[s * 4 + i], Rv, c0, c1, fr);
        const float v0 = vectors[((size_t)i * Rv + c0) * F + f], v1 = vectors[((size_t)i * Rv + c1) * F + f];
        sv[i] = v0 + fr * (v1 - v0);
    }
    const float r = ((__half2float(f_xyz[index]) * sv[3] + __half2float(f_xyt[index]) * sv[2]) +
                     __half2float(f_yzt[index]) * sv[0]) + __half2float(f_xzt[index]) * sv[1];
    out[index] = __float2half(r);
}

__global__ __launch_bounds__(256) void k_compose_bwd(
    const __half* __restrict__ f_xyz, const __half* __restrict__ f_xyt, const __half* __restrict__ f_yzt,
    const __half* __restrict__ f_xzt, const float* __restrict__ vectors, const float* __restrict__ xyzt,
    const __half* __restrict__ d_out, int64_t n, int F, int Rv, __half* __restrict__ d_xyz, __half* __restrict__ d_xyt,
    __half* __restrict__ d_yzt, __half* __restrict__ d_xzt, float* __restrict__ d_vectors)
{
    const int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= n * F) return;
    const int f = (int)(index % F);
    const int64_t s = index / F;
    const float feats[4] = {__half2float(f_yzt[index]), __half2float(f_xzt[index]), __half2float(f_xyt[index]),
                            __half2float(f_xyz[index])};
    const float dy = __half2float(d_out[index]);
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c0, c1; float fr;
        hrf_vec_tap(xyzt[s * 4 + i], Rv, c0, c1, fr);
        const float v0 = vectors[((size_t)i * Rv + c0) * F + f], v1 = vectors[((size_t)i * Rv + c1) * F + f];
        sv[i] = v0 + fr * (v1 - v0);
        const float dval = feats[i] * dy;
        unsafeAtomicAdd(&d_vectors[((size_t)i * Rv + c0) * F + f], dval * (1.0f - fr));
        unsafeAtomicAdd(&d_vectors[((size_t)i * Rv + c1) * F + f], dval * fr);
    }
    // __float2half(sampled_vectors[i] * d_output) (tensor_composition.cu:112-115): the fp32 product, then the conversion --
    // hrf_boundary_half keeps the compiler from fusing the two into v_fma_mixlo_f16, which rounds the exact product once
    d_xyz[index] = hrf_boundary_half(sv[3], dy);
    d_xyt[index] = hrf_boundary_half(sv[2], dy);
    d_yzt[index] = hrf_boundary_half(sv[0], dy);
    d_xzt[index] = hrf_boundary_half(sv[1], dy);
}

extern "C" int hrf_compose_fwd(const void* xyz_f, const void* xyt_f, const void* yzt_f, const void* xzt_f,
                               const float* vectors, const float* xyzt, int64_t n, int feature_dim, int vec_res,
                               void* out_f, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(xyz_f && xyt_f && yzt_f && xzt_f && vectors && xyzt && out_f, "NULL argument");
    HRF_CHECK_ARG(feature_dim > 0 && vec_res > 1, "bad sizes");
    hipLaunchKernelGGL(k_compose_fwd, dim3(hrf_blocks(n * feature_dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)xyz_f, (const __half*)xyt_f, (const __half*)yzt_f, (const __half*)xzt_f, vectors,
                       xyzt, n, feature_dim, vec_res, (__half*)out_f);
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_compose_bwd(const void* xyz_f, const void* xyt_f, const void* yzt_f, const void* xzt_f,
                               const float* vectors, const float* xyzt, const void* d_out, int64_t n, int feature_dim,
                               int vec_res, void* d_xyz, void* d_xyt, void* d_yzt, void* d_xzt, float* d_vectors,
                               hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(xyz_f && xyt_f && yzt_f && xzt_f && vectors && xyzt && d_out, "NULL input");
    HRF_CHECK_ARG(d_xyz && d_xyt && d_yzt && d_xzt && d_vectors, "NULL output");
    hipLaunchKernelGGL(k_compose_bwd, dim3(hrf_blocks(n * feature_dim, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)xyz_f, (const __half*)xyt_f, (const __half*)yzt_f, (const __half*)xzt_f, vectors,
                       xyzt, (const __half*)d_out, n, feature_dim, vec_res, (__half*)d_xyz, (__half*)d_xyt,
                       (__half*)d_yzt, (__half*)d_xzt, d_vectors);
    HRF_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// One stand-alone tcnn HashGrid encoding (tcnn.Encoding(n_input_dims=3, {"otype": "HashGrid", ...}) as the reference
// instantiates it four times per Decomposition4D, decomposition4d.py:79-122), for code written against tinycudann's
// module surface (humanrf_amd.compat.tinycudann). The training path never runs these: it uses the fused kernels above.
// Thread = (sample, level); forward writes __half features level-major inside the row (feature l*2+f), backward
// scatters with fp32 atomics (tcnn: __half2 atomics) after un-scaling by grad_scale.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hashgrid_fwd(const float* __restrict__ x, const __half2* __restrict__ table,
                                                      const hrf_segment_meta* __restrict__ meta, int64_t n,
                                                      __half2* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int L = (int)meta->n_levels;
    if (idx >= n * L) return;
    const int64_t s = idx / L;
    const int l = (int)(idx % L);
    const hrf_level_meta lv = meta->levels[l];
    float f0, f1;
    enc_gather(table + lv.offset, x[s * 3 + 0], x[s * 3 + 1], x[s * 3 + 2], lv, f0, f1);
    out[s * L + l] = __floats2half2_rn(f0, f1);
}

template <bool kHalf>
__global__ __launch_bounds__(256) void k_hashgrid_bwd(const float* __restrict__ x, const hrf_segment_meta* __restrict__ meta,
                                                      int64_t n, const void* __restrict__ d_out, float inv_scale,
                                                      float* __restrict__ d_table)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int L = (int)meta->n_levels;
    if (idx >= n * L) return;
    const int64_t s = idx / L;
    const int l = (int)(idx % L);
    const hrf_level_meta lv = meta->levels[l];
    float2 dy;
    if (kHalf) dy = __half22float2(((const __half2*)d_out)[s * L + l]);
    else dy = ((const float2*)d_out)[s * L + l];
    dy.x *= inv_scale; dy.y *= inv_scale;
    Corner8 cr;
    enc_corners(x[s * 3 + 0], x[s * 3 + 1], x[s * 3 + 2], lv, cr);
    float* tg = d_table + 2 * (size_t)lv.offset;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k], cr.w[k] * dy.x);
        unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k] + 1, cr.w[k] * dy.y);
    }
}

extern "C" int hrf_hashgrid_fwd(const float* x, const void* table, const hrf_segment_meta* meta, int n_levels, int64_t n,
                                void* out_features, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(x && table && meta && out_features, "NULL argument");
    HRF_CHECK_ARG(n_levels > 0 && n_levels <= HRF_MAX_LEVELS, "bad level count");
    hipLaunchKernelGGL(k_hashgrid_fwd, dim3(hrf_blocks(n * n_levels, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (const __half2*)table, meta, n, (__half2*)out_features);
    HRF_CHECK_LAUNCH();
    return 0;
}

extern "C" int hrf_hashgrid_bwd(const float* x, const hrf_segment_meta* meta, int n_levels, int64_t n, const void* d_features,
                                int d_features_fp32, float grad_scale, float* d_table, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(x && meta && d_features && d_table, "NULL argument");
    HRF_CHECK_ARG(n_levels > 0 && n_levels <= HRF_MAX_LEVELS && grad_scale > 0.0f, "bad arguments");
    const dim3 grid(hrf_blocks(n * n_levels, 256)), blk(256);
    if (d_features_fp32)
        hipLaunchKernelGGL(k_hashgrid_bwd<false>, grid, blk, 0, (hipStream_t)stream, x, meta, n, d_features, 1.0f / grad_scale, d_table);
    else
        hipLaunchKernelGGL(k_hashgrid_bwd<true>, grid, blk, 0, (hipStream_t)stream, x, meta, n, d_features, 1.0f / grad_scale, d_table);
    HRF_CHECK_LAUNCH();
    return 0;
}
