// Table-gradient scatter of the fused training path WITHOUT memory-side atomics (gfx950).
//
// Replaces tcnn's kernel_grid_backward x4 + the feature part of compose_tensors_backward
// (humanrf/scene_representation/decomposition4d.py:79-122, native/tensor_composition.cu:85-117) like
// k_encode4d_bwd_tables_lm in encode.hip does, with the same sums in a different order.
//
// Why. On MI355X a device-scope global atomic is executed behind the XCD's L2 (the eight L2s are not coherent with each
// other) and the chip retires ~21 G atomic REQUESTS per second whatever their width or footprint
// (profiles/r01_microbench_atomic_rates.txt). The level-major kernel had been brought down to 58 requests per rendered
// sample -- every ray opens new cells on every level, and rays of a training batch do not share fine cells -- and sat at
// 0.86 of that ceiling: 2.04 ms of a 4.9 ms step (profiles/r02_microbench_scatter_probe.txt). LDS atomics and streaming
// stores have no such ceiling, so the scatter is re-expressed as a two-kernel radix partition + local accumulation:
//
//   k_scatter_emit        walks the samples along their rays (thread = one run of 16 consecutive samples of ONE encoding
//                         and level; the eight corner gradients of the current cell live in registers and follow the walk
//                         from cell to cell), and whenever a corner leaves the walk its (entry index, d_f0, d_f1) record
//                         is appended to the queue of the 8192-entry chunk of the level table the entry lies in. Queues
//                         are private to a (tile of 1024 samples, level, encoding): the slot comes from an LDS counter,
//                         no global atomic is issued and nothing is ordered between workgroups.
//   k_scatter_accumulate  one workgroup per (segment, encoding, level, chunk): 64 KB of fp32 accumulators in LDS,
//                         streams the chunk's queues of all tiles of that segment (ds_add_f32), and adds the chunk to
//                         d_tables with coalesced 64-byte requests.
//
// Record traffic is ~100 records x 12 B per rendered sample, written once and read once: ~1.5 GB per step against the
// 37 M atomic requests it replaces. Queues have a fixed capacity; a record that does not fit (and every sample whose
// temporal segment differs from the first sample's of its tile) takes the direct atomic path, so the result never
// depends on the capacities. Level tables above 8 chunks (65 536 entries) are not handled here: hrf_encode4d_bwd's
// level-major kernel serves those models.
#include "encode_common.h"

#define SB_TS 1024                 // samples per tile (= workgroup of the emit kernel)
#define SB_RL 16                   // consecutive samples walked by one thread
#define SB_RUNS (SB_TS / SB_RL)    // 64 = one wavefront of runs per encoding
#define SB_PAD (SB_RUNS + 1)       // LDS row pitch: sample (run r, step k) sits at k * SB_PAD + r
#define SB_CHUNK_LOG2 13
#define SB_CHUNK (1 << SB_CHUNK_LOG2)   // table entries per accumulate workgroup (2 x fp32 each = 64 KB of LDS)
#define SB_QMAX 8                  // chunks per level table at most
#define SB_CT 8192                 // record capacity per (tile, level, encoding): 8 records per sample
#define SB_LEVELS HRF_MAX_LEVELS

struct SbRec {
    uint32_t key;   // entry index inside the level table
    float a0, a1;   // gradient of the entry's two features
};

// log2 of the number of queues the records of one (tile, level, encoding) are split over: chunks of the level table,
// rounded up to a power of two
__host__ __device__ static inline int sb_queue_shift(uint32_t level_size)
{
    const uint32_t chunks = (level_size + SB_CHUNK - 1) >> SB_CHUNK_LOG2;
    int s = 0;
    while ((1u << s) < chunks) ++s;
    return s;
}

struct SbWorkspace {
    SbRec* recs;             // [tile][level][encoding][SB_CT]
    uint32_t* counts;        // [level][encoding][queue][tile_cap]
    int32_t* tile_seg;       // [tile_cap] segment the tile's queues belong to
    uint32_t* seg_epoch;     // [num_segments] == epoch when the segment has queued records in this call
    int64_t tile_cap;
};

static inline size_t sb_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sb_layout(int64_t n_samples_max, int num_segments, char* base, SbWorkspace* ws)
{
    const int64_t tiles = (n_samples_max + SB_TS - 1) / SB_TS;
    size_t off = 0;
    const size_t b_seg = sb_align((size_t)num_segments * 4);
    const size_t b_tile = sb_align((size_t)tiles * 4);
    const size_t b_cnt = sb_align((size_t)SB_LEVELS * 4 * SB_QMAX * tiles * 4);
    const size_t b_rec = sb_align((size_t)tiles * SB_LEVELS * 4 * SB_CT * sizeof(SbRec));
    if (ws) {
        ws->seg_epoch = (uint32_t*)(base + off);
        ws->tile_seg = (int32_t*)(base + off + b_seg);
        ws->counts = (uint32_t*)(base + off + b_seg + b_tile);
        ws->recs = (SbRec*)(base + off + b_seg + b_tile + b_cnt);
        ws->tile_cap = tiles;
    }
    off += b_seg + b_tile + b_cnt + b_rec;
    return off;
}

extern "C" size_t hrf_scatter_workspace_bytes(int64_t n_samples_max, int num_segments)
{
    if (n_samples_max <= 0 || num_segments <= 0) return 0;
    return sb_layout(n_samples_max, num_segments, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// emit
// ------------------------------------------------------------------------------------------------
#define SB_LG 4                                // levels walked by one workgroup, one after the other: l = group + 4 * i
#define SB_SPT (SB_TS / 256)                   // samples per thread in the per-sample phases

// What a sample contributes that does not depend on the level: coordinates, segment, and per 1-D vector the element
// offsets of its two taps and the fraction between them (tensor_composition.cu:37-45).
struct SbSample {
    float qc[4];
    int sg;                 // temporal segment; -1: beyond the end of the batch
    uint32_t o0[4];         // element offset of the first tap's row inside `vectors` (feature 0); bit 0: the second tap
    float fr[4];            // is the same row (clamped at the end of the vector), otherwise the next one
};
// The loads of one (sample, level): issued one level ahead, consumed when the level's LDS tile is filled.
struct SbLevelIn {
    float2 v0[4], v1[4], dy;
};

// A sample whose temporal segment is not the tile's: its 4 x 8 corners go to memory directly (rare: the batch is laid out
// by frame, a handful of tiles per step hold a segment boundary).
// (arguments by value: an array handed over by address would pin the caller's copy to the stack on the hot path too)
__device__ __noinline__ void sb_direct(float q0, float q1, float q2, float q3, float s00, float s01, float s10, float s11,
                                       float s20, float s21, float s30, float s31, float dyx, float dyy, float inv_scale,
                                       const hrf_segment_meta* __restrict__ sm, int l, float* __restrict__ d_tables)
{
    const hrf_level_meta slv = sm->levels[l];
    const float qc[4] = {q0, q1, q2, q3};
    const float sv[4][2] = {{s00, s01}, {s10, s11}, {s20, s21}, {s30, s31}};
    const float2 dy = make_float2(dyx, dyy);
    const int ax[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
    const int pv[4] = {3, 2, 0, 1};
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
        Corner8 cr;
        enc_corners(qc[ax[ee][0]], qc[ax[ee][1]], qc[ax[ee][2]], slv, cr);
        const float g0 = sv[pv[ee]][0] * dy.x * inv_scale, g1 = sv[pv[ee]][1] * dy.y * inv_scale;
        float* tg = d_tables + 2 * (sm->table_offset + (size_t)ee * sm->entries + slv.offset);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k], cr.w[k] * g0);
            unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k] + 1, cr.w[k] * g1);
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_scatter_emit(
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const float* __restrict__ vectors,
    const hrf_segment_meta* __restrict__ segs, int vec_res, int64_t n, const float* __restrict__ dY_lm, float inv_scale,
    float* __restrict__ d_tables, SbWorkspace ws, uint32_t epoch)
{
    // Per-sample quantities of ONE level, computed with one thread per sample (they do not depend on the walk): cell
    // coordinate and fraction per axis, and per (encoding, feature) the upstream gradient of the encoding's output
    // d_feat_e[f] = v[pair(e)][f] * dY[f] / grad_scale (tensor_composition.cu:112-115).
    __shared__ uint32_t s_cell[2][SB_RL * SB_PAD];     // [0] = cell_x | cell_y << 16, [1] = cell_z | cell_t << 16
    __shared__ float s_w[4][SB_RL * SB_PAD];
    __shared__ float s_g[4][2][SB_RL * SB_PAD];
    __shared__ uint32_t s_cnt[4][SB_QMAX];
    const int tid = threadIdx.x, lane = tid & 63, e = tid >> 6;
    // The 58 KB tile allows two workgroups per CU (two wavefronts per SIMD), too few to hide memory latency by switching:
    // a workgroup therefore walks FOUR levels of its tile one after the other (l = group, group + 4, ...: every workgroup
    // gets the same mix of coarse levels, where the walk dominates, and fine ones, a record per corner and sample) and
    // the global loads of the next level (vector taps, dY) are in flight while the current one is walked.
    const int group = (int)(blockIdx.x % SB_LG);
    const int64_t tile = blockIdx.x / SB_LG;
    const int64_t base = tile * SB_TS;
    const int n_here = (int)min((int64_t)SB_TS, n - base);
    const int tseg = segment ? segment[base] : 0;
    const hrf_segment_meta* tsm = segs + tseg;
    const int tile_levels = (int)tsm->n_levels;
    // encoding e: axes (a,b,c) = xyz, xyt, yzt, xzt; pairs with vector {3, 2, 0, 1}[e] (tensor_composition.cu:47-54)
    const int pv[4] = {3, 2, 0, 1};

    SbSample sm[SB_SPT];
#pragma unroll
    for (int it = 0; it < SB_SPT; ++it) {
        const int sl = it * 256 + tid;
        sm[it].sg = -1;
        if (sl < n_here) {
            const float4 q4 = ((const float4*)xyzt)[base + sl];
            const int sg = segment ? segment[base + sl] : 0;
            sm[it].sg = sg;
            sm[it].qc[0] = q4.x; sm[it].qc[1] = q4.y; sm[it].qc[2] = q4.z; sm[it].qc[3] = q4.w;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int c0, c1;
                hrf_vec_tap(sm[it].qc[v], vec_res, c0, c1, sm[it].fr[v]);
                const uint32_t row = (uint32_t)(sg * 4 + v) * (uint32_t)vec_res;
                sm[it].o0[v] = ((row + (uint32_t)c0) * ENC_F) | (c1 == c0 ? 1u : 0u);
            }
        }
    }
    auto prefetch = [&](int l, SbLevelIn* in) {
#pragma unroll
        for (int it = 0; it < SB_SPT; ++it) {
            if (sm[it].sg < 0 || l >= (int)segs[sm[it].sg].n_levels) continue;
            const float* vb = vectors + 2 * l;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t o = sm[it].o0[v] & ~1u;
                in[it].v0[v] = *(const float2*)(vb + o);
                in[it].v1[v] = *(const float2*)(vb + o + ((sm[it].o0[v] & 1u) ? 0u : (uint32_t)ENC_F));
            }
            in[it].dy = *(const float2*)(dY_lm + ((size_t)l * n + base + it * 256 + tid) * 2);
        }
    };
    SbLevelIn cur[SB_SPT];
    prefetch(group, cur);

#pragma unroll 1
    for (int li = 0; li < SB_LEVELS / SB_LG; ++li) {
        const int l = group + SB_LG * li;
        const bool tile_has_level = l < tile_levels;
        hrf_level_meta lv;
        lv.scale = 0; lv.res = 1; lv.size = 1; lv.offset = 0; lv.hashed = 0;
        if (tile_has_level) lv = tsm->levels[l];
        // ---- this level's LDS tile
#pragma unroll
        for (int it = 0; it < SB_SPT; ++it) {
            const int sl = it * 256 + tid;
            const int p = (sl % SB_RL) * SB_PAD + sl / SB_RL;     // LDS slot of the sample: (step k, run r)
            uint32_t c01 = 0xFFFFFFFFu, c23 = 0u;                  // 0xFFFFFFFF: nothing to walk for this sample
            const int sg = sm[it].sg;
            if (sg >= 0 && l < (int)segs[sg].n_levels) {
                float sv[4][2];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    sv[v][0] = cur[it].v0[v].x + sm[it].fr[v] * (cur[it].v1[v].x - cur[it].v0[v].x);
                    sv[v][1] = cur[it].v0[v].y + sm[it].fr[v] * (cur[it].v1[v].y - cur[it].v0[v].y);
                }
                if (sg == tseg) {
                    uint32_t ci[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float pp = fmaf(sm[it].qc[v], lv.scale, 0.5f);
                        const float fl = floorf(pp);
                        ci[v] = (uint32_t)(int)fl;
                        s_w[v][p] = pp - fl;
                    }
                    c01 = (ci[0] & 0xFFFFu) | (ci[1] << 16);
                    c23 = (ci[2] & 0xFFFFu) | (ci[3] << 16);
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) {
                        s_g[ee][0][p] = sv[pv[ee]][0] * cur[it].dy.x * inv_scale;
                        s_g[ee][1][p] = sv[pv[ee]][1] * cur[it].dy.y * inv_scale;
                    }
                } else {
                    sb_direct(sm[it].qc[0], sm[it].qc[1], sm[it].qc[2], sm[it].qc[3], sv[0][0], sv[0][1], sv[1][0], sv[1][1],
                              sv[2][0], sv[2][1], sv[3][0], sv[3][1], cur[it].dy.x, cur[it].dy.y, inv_scale, segs + sg, l,
                              d_tables);
                }
            }
            s_cell[0][p] = c01;
            s_cell[1][p] = c23;
        }
        if (tid < 4 * SB_QMAX) s_cnt[tid / SB_QMAX][tid % SB_QMAX] = 0u;
        __syncthreads();
        if (li + 1 < SB_LEVELS / SB_LG) prefetch(l + SB_LG, cur);     // in flight during the walk below

        const int qshift = sb_queue_shift(lv.size);
        const int sub_shift = 13 - qshift;                           // SB_CT = 2^13 records, split over 2^qshift queues
        if (tile_has_level) {   // (wave-uniform)
            const uint32_t sub_cap = 1u << sub_shift;
            char* rbase = (char*)(ws.recs + (((size_t)tile * SB_LEVELS + l) * 4 + e) * SB_CT);
            float* tg = d_tables + 2 * (tsm->table_offset + (size_t)e * tsm->entries + lv.offset);
            uint32_t* cnt = s_cnt[e];
            const uint32_t lv_res = lv.res, lv_size = lv.size, lv_hashed = lv.hashed;
            const float* wA = s_w[(e == 2) ? 1 : 0];
            const float* wB = s_w[(e <= 1) ? 1 : 2];
            const float* wC = s_w[(e == 0) ? 2 : 3];
            const float* g0p = s_g[e][0];
            const float* g1p = s_g[e][1];
            float acc[8][2];
            uint32_t key[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j][0] = 0.0f; acc[j][1] = 0.0f; key[j] = 0u; }

            // Append the corners selected by `mask` to the queues of their chunks: slots from the LDS counters first (all
            // eight in flight together), then the 12-byte records; a corner whose queue is full goes to memory directly.
            auto emit = [&](uint32_t mask) {
                uint32_t slot[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    slot[j] = 0u;
                    if (mask & (1u << j)) slot[j] = atomicAdd(&cnt[key[j] >> SB_CHUNK_LOG2], 1u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (mask & (1u << j)) {
                        if (__builtin_expect(slot[j] < sub_cap, 1)) {
                            const uint32_t idx = ((key[j] >> SB_CHUNK_LOG2) << sub_shift) + slot[j];
                            SbRec r; r.key = key[j]; r.a0 = acc[j][0]; r.a1 = acc[j][1];
                            *(SbRec*)(rbase + idx * (uint32_t)sizeof(SbRec)) = r;
                        } else {
                            unsafeAtomicAdd(tg + 2 * (size_t)key[j], acc[j][0]);
                            unsafeAtomicAdd(tg + 2 * (size_t)key[j] + 1, acc[j][1]);
                        }
                    }
                }
            };

            uint32_t pa = 0, pb = 0, pc = 0;
            bool have = false;
#pragma unroll 1
            for (int k = 0; k < SB_RL; ++k) {
                const int p = k * SB_PAD + lane;
                const uint32_t c01 = s_cell[0][p], c23 = s_cell[1][p];
                if (c01 == 0xFFFFFFFFu) continue;
                uint32_t ia, ib, ic;
                if (e == 0)      { ia = c01 & 0xFFFFu; ib = c01 >> 16;     ic = c23 & 0xFFFFu; }
                else if (e == 1) { ia = c01 & 0xFFFFu; ib = c01 >> 16;     ic = c23 >> 16; }
                else if (e == 2) { ia = c01 >> 16;     ib = c23 & 0xFFFFu; ic = c23 >> 16; }
                else             { ia = c01 & 0xFFFFu; ib = c23 & 0xFFFFu; ic = c23 >> 16; }
                const float wa = wA[p], wb = wB[p], wc = wC[p];
                const float g0 = g0p[p], g1 = g1p[p];
                if (!have || ia != pa || ib != pb || ic != pc) {
                    // Cell change. Neighbouring cells share corners: a corner of the old cell that is also a corner of the
                    // new one keeps its running sum (it moves to the register of its new role); the others leave the walk.
                    const int mx = (int)(ia - pa), my = (int)(ib - pb), mz = (int)(ic - pc);
                    const bool adjacent = have && (unsigned)(mx + 1) <= 2u && (unsigned)(my + 1) <= 2u && (unsigned)(mz + 1) <= 2u;
                    if (have) {
                        // old role c on an axis survives iff c - m is 0 or 1: m == 0, or m == 2c - 1
                        const bool kx[2] = {adjacent && mx <= 0, adjacent && mx >= 0};
                        const bool ky[2] = {my <= 0, my >= 0};
                        const bool kz[2] = {mz <= 0, mz >= 0};
                        uint32_t mask = 0u;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bool kept = kx[j & 1] && ky[(j >> 1) & 1] && kz[(j >> 2) & 1];
                            if (!kept && (acc[j][0] != 0.0f || acc[j][1] != 0.0f)) mask |= 1u << j;
                        }
                        if (mask) emit(mask);
                    }
                    // new role c continues old role c + m (when that is 0 or 1), axis by axis; vacated roles start at zero
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {   // x: pairs (j, j+1)
                            const float o0 = acc[j][f], o1 = acc[j + 1][f];
                            acc[j][f] = !adjacent ? 0.0f : (mx == 0 ? o0 : (mx == 1 ? o1 : 0.0f));
                            acc[j + 1][f] = !adjacent ? 0.0f : (mx == 0 ? o1 : (mx == -1 ? o0 : 0.0f));
                        }
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {   // y: pairs (j, j+2)
                            const int j = (jj & 1) | ((jj & 2) << 1);
                            const float o0 = acc[j][f], o1 = acc[j + 2][f];
                            acc[j][f] = my == 0 ? o0 : (my == 1 ? o1 : 0.0f);
                            acc[j + 2][f] = my == 0 ? o1 : (my == -1 ? o0 : 0.0f);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {      // z: pairs (j, j+4)
                            const float o0 = acc[j][f], o1 = acc[j + 4][f];
                            acc[j][f] = mz == 0 ? o0 : (mz == 1 ? o1 : 0.0f);
                            acc[j + 4][f] = mz == 0 ? o1 : (mz == -1 ? o0 : 0.0f);
                        }
                    }
                    // entry indices of the new cell: enc_corners' (tcnn grid_index) -- mask on hashed levels (their size
                    // is a power of two), the stride form on dense ones, which wraps only for far corners of the last cells
                    if (lv_hashed) {
                        const uint32_t hb0 = ib * 2654435761u, hb1 = (ib + 1u) * 2654435761u;
                        const uint32_t hc0 = ic * 805459861u, hc1 = (ic + 1u) * 805459861u;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            key[j] = ((ia + (j & 1)) ^ ((j & 2) ? hb1 : hb0) ^ ((j & 4) ? hc1 : hc0)) & (lv_size - 1u);
                    } else {
                        const uint32_t rr = lv_res * lv_res;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            uint32_t i = (ia + (j & 1)) + (ib + ((j >> 1) & 1)) * lv_res + (ic + ((j >> 2) & 1)) * rr;
                            if (i >= lv_size) { i -= lv_size; if (i >= lv_size) i %= lv_size; }
                            key[j] = i;
                        }
                    }
                    pa = ia; pb = ib; pc = ic; have = true;
                }
                // corner weights as enc_corners forms them: ((1 * wx) * wy) * wz
                const float wx[2] = {1.0f - wa, wa}, wy[2] = {1.0f - wb, wb}, wz[2] = {1.0f - wc, wc};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float w = 1.0f * wx[j & 1];
                    w *= wy[(j >> 1) & 1];
                    w *= wz[(j >> 2) & 1];
                    acc[j][0] = fmaf(w, g0, acc[j][0]);
                    acc[j][1] = fmaf(w, g1, acc[j][1]);
                }
            }
            if (have) {
                uint32_t mask = 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (acc[j][0] != 0.0f || acc[j][1] != 0.0f) mask |= 1u << j;
                if (mask) emit(mask);
            }
        }
        __syncthreads();
        if (tid < 4 * SB_QMAX) {
            const int ee = tid / SB_QMAX, q = tid % SB_QMAX;
            const uint32_t sub_cap = tile_has_level ? (1u << sub_shift) : 0u;
            ws.counts[(((size_t)l * 4 + ee) * SB_QMAX + q) * ws.tile_cap + tile] = min(s_cnt[ee][q], sub_cap);
        }
    }
    if (tid == 0 && group == 0) {
        ws.tile_seg[tile] = tseg;
        ws.seg_epoch[tseg] = epoch;
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate
// ------------------------------------------------------------------------------------------------
#define SB_ACC_THREADS 512
#define SB_ACC_UNROLL 6     // records per lane in flight: one pass covers queues of up to 384 records
template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_scatter_accumulate(
    const hrf_segment_meta* __restrict__ segs, SbWorkspace ws, uint32_t epoch, int64_t n_tiles, float* __restrict__ d_tables)
{
    __shared__ float s_acc[2 * SB_CHUNK];
    const int q = (int)(blockIdx.x % SB_QMAX);
    const int e = (int)((blockIdx.x / SB_QMAX) % 4);
    const int l = (int)((blockIdx.x / (SB_QMAX * 4)) % SB_LEVELS);
    const int seg = (int)(blockIdx.x / (SB_QMAX * 4 * SB_LEVELS));
    if (ws.seg_epoch[seg] != epoch) return;
    if (l >= (int)segs[seg].n_levels) return;
    const hrf_level_meta lv = segs[seg].levels[l];
    const int qshift = sb_queue_shift(lv.size);
    if (q >= (1 << qshift) || ((uint32_t)q << SB_CHUNK_LOG2) >= lv.size) return;
    const uint32_t sub_cap = (uint32_t)SB_CT >> qshift;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kThreads / 64;
    for (int i = tid; i < 2 * SB_CHUNK; i += kThreads) s_acc[i] = 0.0f;
    __syncthreads();
    const uint32_t* cnts = ws.counts + (((size_t)l * 4 + e) * SB_QMAX + q) * ws.tile_cap;
    const uint32_t kbase = (uint32_t)q << SB_CHUNK_LOG2;
    // One wavefront per tile queue. The queue's owner and length are fetched one tile ahead (two dependent memory round
    // trips per tile otherwise stand in front of the one that streams records), and a lane keeps SB_ACC_UNROLL records
    // in flight.
    int64_t t = wave;
    int own = (t < n_tiles) ? ws.tile_seg[t] : -1;
    int cnt = (t < n_tiles) ? (int)cnts[t] : 0;
#pragma unroll 1
    while (t < n_tiles) {
        const int64_t tn = t + kWaves;
        const int own_n = (tn < n_tiles) ? ws.tile_seg[tn] : -1;
        const int cnt_n = (tn < n_tiles) ? (int)cnts[tn] : 0;
        if (own == seg && cnt > 0) {
            const SbRec* src = ws.recs + (((size_t)t * SB_LEVELS + l) * 4 + e) * SB_CT + (size_t)q * sub_cap;
#pragma unroll 1
            for (int i0 = 0; i0 < cnt; i0 += 64 * SB_ACC_UNROLL) {
                SbRec r[SB_ACC_UNROLL];
#pragma unroll
                for (int u = 0; u < SB_ACC_UNROLL; ++u) {
                    const int i = i0 + u * 64 + lane;
                    r[u].key = kbase; r[u].a0 = 0.0f; r[u].a1 = 0.0f;
                    if (i < cnt) r[u] = src[i];
                }
#pragma unroll
                for (int u = 0; u < SB_ACC_UNROLL; ++u) {
                    const int i = i0 + u * 64 + lane;
                    if (i < cnt) {
                        const uint32_t k = (r[u].key - kbase) & (SB_CHUNK - 1);
                        atomicAdd(&s_acc[2 * k], r[u].a0);
                        atomicAdd(&s_acc[2 * k + 1], r[u].a1);
                    }
                }
            }
        }
        t = tn; own = own_n; cnt = cnt_n;
    }
    __syncthreads();
    const uint32_t n_here = min((uint32_t)SB_CHUNK, lv.size - kbase);
    float* tg = d_tables + 2 * (segs[seg].table_offset + (size_t)e * segs[seg].entries + lv.offset + kbase);
    for (uint32_t i = tid; i < 2 * n_here; i += kThreads) {
        const float v = s_acc[i];
        if (v != 0.0f) unsafeAtomicAdd(tg + i, v);   // 16 lanes = one 64-byte request; the direct path may add to it too
    }
}

extern "C" int hrf_encode4d_bwd_tables_binned(const float* xyzt, const int32_t* segment, const float* vectors,
                                              const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                              const float* d_features_lm, float grad_scale, float* d_tables,
                                              void* workspace, int64_t workspace_samples, uint32_t epoch,
                                              int max_level_entries, int deterministic, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(xyzt && vectors && segments && d_features_lm && d_tables && workspace, "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && vec_res > 1 && grad_scale > 0.0f, "bad arguments");
    HRF_CHECK_ARG(epoch != 0u, "epoch must be non-zero and differ from the previous call's on this workspace");
    HRF_CHECK_ARG(max_level_entries > 0 && max_level_entries <= SB_QMAX * SB_CHUNK,
                  "level tables above 65536 entries are served by hrf_encode4d_bwd (d_features_mode 2)");
    HRF_CHECK_ARG(n <= workspace_samples, "batch larger than the workspace was sized for (hrf_scatter_workspace_bytes)");
    SbWorkspace ws;
    const int64_t tiles = (n + SB_TS - 1) / SB_TS;
    sb_layout(workspace_samples, num_segments, (char*)workspace, &ws);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scatter_emit, dim3((unsigned)(tiles * SB_LG)), dim3(256), 0, st, xyzt, segment, vectors, segments,
                       vec_res, n, d_features_lm, 1.0f / grad_scale, d_tables, ws, epoch);
    HRF_CHECK_LAUNCH();
    const dim3 grid((unsigned)(num_segments * SB_LEVELS * 4 * SB_QMAX));
    if (deterministic)   // one wavefront per chunk: records are added in queue order
        hipLaunchKernelGGL(k_scatter_accumulate<64>, grid, dim3(64), 0, st, segments, ws, epoch, tiles, d_tables);
    else
        hipLaunchKernelGGL(k_scatter_accumulate<SB_ACC_THREADS>, grid, dim3(SB_ACC_THREADS), 0, st, segments, ws, epoch, tiles, d_tables);
    HRF_CHECK_LAUNCH();
    return 0;
}
