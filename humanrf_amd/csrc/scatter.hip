// Table-gradient scatter of the fused training path WITHOUT memory-side atomics (gfx950).
//
// Replaces tcnn's kernel_grid_backward x4 + the feature part of compose_tensors_backward
// (humanrf/scene_representation/decomposition4d.py:79-122, native/tensor_composition.cu:85-117) like
// k_encode4d_bwd_tables_lm in encode.hip does, with the same sums in a different order.
//
// Why. On MI355X a global atomic is executed behind the XCD's L2 (the eight L2s are not coherent with each other) and
// the chip retires 21-27 G atomic REQUESTS per second whatever their width, type (fp32 / u32 / u64 / fp64) or footprint
// (profiles/r01_microbench_atomic_rates.txt, r03_microbench_atomic_types.txt). The level-major kernel had been brought
// down to 58 requests per rendered sample -- every ray opens new cells on every level, and rays of a training batch do
// not share fine cells -- and sat at 0.86 of that ceiling: 2.04 ms of a 4.9 ms step (r02_microbench_scatter_probe.txt).
// Streaming stores and INTEGER LDS atomics have no such ceiling (ds_add_u64: 14 cycles per wavefront instruction;
// ds_add_f32: 194 -- the LDS has no fp32 adder worth the name, r03_microbench_atomic_types.txt), so the scatter is
// re-expressed as a radix partition + local accumulation:
//
//   k_scatter_tiles       cuts the batch (sorted by temporal segment: the collector lays it out by frame) into tiles of
//                         up to 1024 samples that never straddle a segment boundary.
//   k_scatter_emit        walks the samples along their rays (thread = one run of 8 consecutive samples of ONE encoding
//                         and level; the gradients of the eight corners of the current cell live in registers, each in
//                         the slot of its coordinate PARITIES, so a corner shared with the next cell stays where it is),
//                         and whenever a corner leaves the walk its (entry index, d_f0, d_f1) record is appended to the
//                         queue of the 8192-entry chunk of the level table the entry lies in. Queues are private to a
//                         (tile, level, encoding): the slot comes from an LDS counter, no global atomic is issued.
//   k_scatter_accumulate  one workgroup per (segment, encoding, level, chunk): 128 KB of 64-bit FIXED-POINT accumulators
//                         in LDS, streams the chunk's queues of the segment's tiles (ds_add_u64), and adds the chunk to
//                         d_tables with coalesced 64-byte requests. The fixed-point unit is chosen per (segment, level,
//                         encoding) from the largest record in its queues (the emit kernel keeps the maxima): 2^-38 of it,
//                         i.e. 38 bits below the largest contribution where tcnn's __half accumulator has 11 bits of
//                         mantissa and 40 binades in all. Integer sums do not depend on the order of the records: the
//                         table gradients are reproducible bit for bit.
//
// Record traffic is ~113 records x 12 B per rendered sample, written once and read once (~1.7 GB per step) against the
// 37 M atomic requests it replaces: 1.0 ms for the two kernels where the atomic kernel took 2.0 (MI355X, default bench regime). Queues have a fixed capacity; a record that does not fit (and every sample whose
// temporal segment is not its tile's, which only happens when the batch is not sorted by segment) takes the direct
// atomic path, so the result never depends on the capacities or on the order of the batch.
//
// Level tables of up to 64 chunks (2^19 entries: the largest table the reference's default log2_hashmap_size = 19 gives a
// 100-frame segment, humanrf.py:106-109) are served. Tables above 8 chunks use an INTERLEAVED chunk map: chunk = bits
// [4, 4 + log2 chunks) of the entry index, i.e. runs of 16 entries (one 128-byte line of d_tables) are dealt out to the
// chunks round-robin. On hashed levels any bits of the index are uniform; on the dense levels that exceed 65 536 entries
// (res^3 <= T: res 42 / 55 at T = 2^18, 73 at 2^19) the high bits are the z slab, and the body of one frame would fill a
// third of the queues three times over while the rest stay empty.
#include "encode_common.h"
#include <type_traits>

#define SB_TS 1024                 // samples per tile (= workgroup of the emit kernel) at most
#ifndef SB_RL
#define SB_RL 8                    // consecutive samples walked by one thread (measured, 16-samples-per-ray regime: 16 -> 1.67 ns per sample
                                   // for the two kernels, 85 records per sample; 8 -> 1.47 ns, 113 records: twice the wavefronts per LDS byte)
#endif
#define SB_RUNS (SB_TS / SB_RL)    // runs per tile = threads per encoding (a multiple of the wavefront)
#define SB_THREADS (4 * SB_RUNS)   // workgroup of the emit kernel: (run, encoding)
#define SB_PAD (SB_RUNS + 1)       // LDS row pitch: sample (run r, step k) sits at k * SB_PAD + r
#ifndef SB_CHUNK_LOG2
#define SB_CHUNK_LOG2 13
#endif
#define SB_CHUNK (1 << SB_CHUNK_LOG2)   // table entries per accumulate workgroup (2 x 64-bit each = 128 KB of LDS at 2^13)
#define SB_QMAX (1 << (19 - SB_CHUNK_LOG2))   // chunks per level table at most (2^19 entries)
#define SB_QCONTIG_LOG2 3          // up to 2^3 chunks: chunk = entry >> 13 (contiguous); above: interleaved by 16-entry lines
#define SB_CT 8192                 // record capacity per (tile, level, encoding): 8 records per sample
#define SB_LEVELS HRF_MAX_LEVELS
#define SB_MAX_SEGMENTS 1024       // temporal segments the tile builder handles
#define SB_FIX_BITS 37             // a record of the largest magnitude 2^e maps to 2^37 .. 2^38: 2^25 such records fit an int64

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

// queue (= chunk) of an entry and its index inside the chunk's accumulators; both maps are bijections between
// [0, chunks * SB_CHUNK) and (queue, local)
__device__ __forceinline__ uint32_t sb_queue_of(uint32_t key, int qshift)
{
    return qshift <= SB_QCONTIG_LOG2 ? (key >> SB_CHUNK_LOG2) : ((key >> 4) & ((1u << qshift) - 1u));
}
__device__ __forceinline__ uint32_t sb_local_of(uint32_t key, int qshift)
{
    return qshift <= SB_QCONTIG_LOG2 ? (key & (SB_CHUNK - 1)) : ((((key >> 4) >> qshift) << 4) | (key & 15u));
}
__device__ __forceinline__ uint32_t sb_entry_of(uint32_t q, uint32_t local, int qshift)
{
    return qshift <= SB_QCONTIG_LOG2 ? ((q << SB_CHUNK_LOG2) | local) : ((((local >> 4) << qshift) | q) << 4) | (local & 15u);
}

struct SbWorkspace {
    SbRec* recs;             // [tile][level][encoding][SB_CT]
    uint32_t* counts;        // [level][encoding][queue][tile_cap]
    uint32_t* maxes;         // [level][encoding][tile_cap] bits of the largest |value| queued by the tile (a float >= 0)
    int32_t* tile_start;     // [tile_cap] first sample of the tile
    int32_t* tile_count;     // [tile_cap] its samples (1 .. SB_TS)
    int32_t* tile_seg;       // [tile_cap] the temporal segment all of them belong to (when the batch is sorted)
    int32_t* seg_tile0;      // [num_segments + 1] first tile of every segment; [num_segments] = number of tiles
    int32_t* seg_list;       // [num_segments + 1] the segments that own tiles, in order; [num_segments] = how many
    uint32_t* seg_amax;      // [num_segments][level][encoding] the largest of `maxes` over the segment's tiles (zeroed by k_scatter_tiles,
                             // raised by the emit kernel: what the accumulate kernel derives its fixed-point unit from)
    int64_t tile_cap;
};

static inline size_t sb_align(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t sb_tile_cap(int64_t n_samples, int num_segments) { return (n_samples + SB_TS - 1) / SB_TS + num_segments; }

static size_t sb_layout(int64_t n_samples_max, int num_segments, char* base, SbWorkspace* ws)
{
    const int64_t tiles = sb_tile_cap(n_samples_max, num_segments);
    const size_t b_seg = sb_align((size_t)(num_segments + 1) * 4);
    const size_t b_tile = sb_align((size_t)tiles * 4);
    const size_t b_head = 2 * b_seg;
    const size_t b_cnt = sb_align((size_t)SB_LEVELS * 4 * SB_QMAX * tiles * 4);
    const size_t b_max = sb_align((size_t)SB_LEVELS * 4 * tiles * 4);
    const size_t b_rec = sb_align((size_t)tiles * SB_LEVELS * 4 * SB_CT * sizeof(SbRec));
    const size_t b_amax = sb_align((size_t)num_segments * SB_LEVELS * 4 * 4);
    if (ws) {
        ws->seg_tile0 = (int32_t*)base;
        ws->seg_list = (int32_t*)(base + b_seg);
        ws->tile_start = (int32_t*)(base + b_head);
        ws->tile_count = (int32_t*)(base + b_head + b_tile);
        ws->tile_seg = (int32_t*)(base + b_head + 2 * b_tile);
        ws->counts = (uint32_t*)(base + b_head + 3 * b_tile);
        ws->maxes = (uint32_t*)(base + b_head + 3 * b_tile + b_cnt);
        ws->recs = (SbRec*)(base + b_head + 3 * b_tile + b_cnt + b_max);
        ws->seg_amax = (uint32_t*)(base + b_head + 3 * b_tile + b_cnt + b_max + b_rec);
        ws->tile_cap = tiles;
    }
    return b_head + 3 * b_tile + b_cnt + b_max + b_rec + b_amax;
}

extern "C" size_t hrf_scatter_workspace_bytes(int64_t n_samples_max, int num_segments)
{
    if (n_samples_max <= 0 || num_segments <= 0) return 0;
    return sb_layout(n_samples_max, num_segments, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// tiles
// ------------------------------------------------------------------------------------------------
// Segment s owns the samples [start[s], start[s+1]), start[s] = first index whose segment id is >= s. For a batch sorted
// by segment these are exactly its samples; for any other batch they still partition [0, n) (the starts are forced to be
// monotone), and the emit kernel sends samples that sit in a foreign tile through the direct path.
__global__ __launch_bounds__(256) void k_scatter_tiles(const int32_t* __restrict__ segment, int64_t n, int num_segments,
                                                        SbWorkspace ws)
{
    __shared__ int32_t s_start[SB_MAX_SEGMENTS + 1];
    __shared__ int32_t s_tile0[SB_MAX_SEGMENTS + 1];
    // lower bound of every segment id in segment[]: G lanes per boundary probe G interior points of the bracket at a time (a
    // (G+1)-ary search: 4 dependent loads for 7 segments and 2^20 samples where the one-lane binary search of rounds 3-5 took 20 --
    // this one-workgroup kernel sits on the step's critical path between the MLP backward and the emit kernel); G = 1 IS the
    // binary search (models with more than 127 segments).
    int G = 1;
    while (G < 64 && 2 * G * (num_segments + 1) <= (int)blockDim.x) G <<= 1;
    const int per_pass = (int)blockDim.x / G, gl = (int)threadIdx.x % G, gi = (int)threadIdx.x / G;
    const int glane0 = ((int)threadIdx.x & 63) - gl;            // first lane of this group inside its wavefront (G divides 64)
    const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    for (int s0 = 0; s0 <= num_segments; s0 += per_pass) {
        const int s = s0 + gi;
        // invariant: ids below `lo` are smaller than s, and hi == n or segment[hi] >= s: the lower bound lies in [lo, hi]
        int32_t lo = 0, hi = (s > 0 && s < num_segments && segment) ? (int32_t)n : 0;
        if (s >= num_segments || (s > 0 && !segment)) lo = hi = (int32_t)n;
        for (;;) {
            const int32_t span = hi - lo;
            if (__all(span <= 0)) break;                          // (wave-uniform: every group of the wavefront is done)
            const int32_t pos = lo + (int32_t)(((int64_t)(gl + 1) * span) / (G + 1));   // < hi; monotone in gl
            const bool below = span > 0 && segment[pos] < s;      // true ... true false ... false over the group
            const int cnt = __popcll((__ballot(below) >> glane0) & gmask);
            // positions of probe cnt-1 (the last one below) and probe cnt (the first one not below), from their owners' lanes
            const int32_t last_below = __shfl(pos, glane0 + max(cnt - 1, 0), 64);
            const int32_t first_not = __shfl(pos, glane0 + min(cnt, G - 1), 64);
            if (span > 0) {
                if (cnt < G) hi = first_not;
                if (cnt > 0) lo = last_below + 1;
            }
        }
        if (gl == 0 && s <= num_segments) s_start[s] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int s = 0; s < num_segments; ++s) {
            if (s_start[s + 1] < s_start[s]) s_start[s + 1] = s_start[s];
            s_tile0[s] = t;
            t += (s_start[s + 1] - s_start[s] + SB_TS - 1) / SB_TS;
        }
        s_tile0[num_segments] = t;
        int32_t present = 0;                       // the segments that own tiles: the accumulate kernel's work list
        for (int s = 0; s < num_segments; ++s)
            if (s_tile0[s + 1] > s_tile0[s]) ws.seg_list[present++] = s;
        ws.seg_list[num_segments] = present;
    }
    __syncthreads();
    for (int s = threadIdx.x; s <= num_segments; s += blockDim.x) ws.seg_tile0[s] = s_tile0[s];
    for (int i = threadIdx.x; i < num_segments * SB_LEVELS * 4; i += blockDim.x) ws.seg_amax[i] = 0u;
    for (int s = 0; s < num_segments; ++s) {
        const int32_t a = s_start[s], len = s_start[s + 1] - a, t0 = s_tile0[s];
        for (int32_t k = threadIdx.x; k * SB_TS < len; k += blockDim.x) {
            ws.tile_start[t0 + k] = a + k * SB_TS;
            ws.tile_count[t0 + k] = min((int32_t)SB_TS, len - k * SB_TS);
            ws.tile_seg[t0 + k] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// emit
// ------------------------------------------------------------------------------------------------
// kHalfG: with the fp16 gradient boundary (gb > 0, the engine's default) the staged upstream gradients are HALF values
// times gb by construction (hrf_through_half), so the tile keeps the halves: 16.5 KB of LDS instead of 33 -- 42 KB per
// workgroup, three workgroups (24 wavefronts) per CU instead of two. Same values: g = float(h) * gb is what
// hrf_through_half returns.
template <bool kHalfG>
__global__ __launch_bounds__(SB_THREADS, (kHalfG ? 6 : 4)) void k_scatter_emit(   // three / two workgroups of 8 wavefronts per CU
    
    const float* __restrict__ xyzt, const int32_t* __restrict__ segment, const float* __restrict__ vectors,
    const hrf_segment_meta* __restrict__ segs, int num_segments, int vec_res, int64_t n, const float* __restrict__ dY_lm,
    float inv_scale, float* __restrict__ d_tables, SbWorkspace ws, float gb)
{
    const float inv_gb = gb > 0.0f ? 1.0f / gb : 0.0f;
    // Per-sample quantities of THIS level, computed once per workgroup with one thread per sample (they do not depend on
    // the walk): cell coordinate and fraction per axis, and per (encoding, feature) the upstream gradient of the
    // encoding's output d_feat_e[f] = v[pair(e)][f] * dY[f] / grad_scale (tensor_composition.cu:112-115).
    __shared__ uint32_t s_cell[2][SB_RL * SB_PAD];     // [0] = cell_x | cell_y << 16, [1] = cell_z | cell_t << 16
    __shared__ float s_w[4][SB_RL * SB_PAD];
    typedef typename std::conditional<kHalfG, __half, float>::type GT;
    __shared__ GT s_g[4][2][SB_RL * SB_PAD];
    __shared__ uint32_t s_cnt[4][SB_QMAX];
    __shared__ uint32_t s_max[4];
    const int tid = threadIdx.x, lane = tid % SB_RUNS, e = tid / SB_RUNS;   // (wave-uniform encoding: SB_RUNS % 64 == 0)
    // consecutive workgroups take the 16 levels of one tile: coarse levels (few cell changes, the walk dominates) and fine
    // levels (a record per corner and sample) run side by side on every CU
    const int l = (int)(blockIdx.x % SB_LEVELS);
    const int64_t tile = blockIdx.x / SB_LEVELS;
    if (tile >= ws.seg_tile0[num_segments]) return;
    const int64_t base = ws.tile_start[tile];
    const int n_here = ws.tile_count[tile];
    const int tseg = ws.tile_seg[tile];
    const bool tile_has_level = l < (int)segs[tseg].n_levels;
    hrf_level_meta lv;
    lv.scale = 0; lv.res = 1; lv.size = 1; lv.offset = 0; lv.hashed = 0;
    if (tile_has_level) lv = segs[tseg].levels[l];
    if (tid < 4 * SB_QMAX) s_cnt[tid / SB_QMAX][tid % SB_QMAX] = 0u;
    if (tid < 4) s_max[tid] = 0u;
    // encoding e: axes (a,b,c) = xyz, xyt, yzt, xzt; pairs with vector {3, 2, 0, 1}[e] (tensor_composition.cu:47-54)
    const int ax[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
    const int pv[4] = {3, 2, 0, 1};

    float gm[4] = {0.0f, 0.0f, 0.0f, 0.0f};                    // largest |upstream gradient| per encoding among this thread's samples
#pragma unroll 1
    for (int it = 0; it < SB_TS / SB_THREADS; ++it) {
        const int sl = it * SB_THREADS + tid;                 // sample of the tile
        const int p = (sl % SB_RL) * SB_PAD + sl / SB_RL;     // its LDS slot: (step k, run r)
        uint32_t c01 = 0xFFFFFFFFu, c23 = 0u;                  // 0xFFFFFFFF: nothing to walk for this sample
        if (sl < n_here) {
            const float4 q4 = ((const float4*)xyzt)[base + sl];
            const int sg = segment ? segment[base + sl] : 0;
            if (l < (int)segs[sg].n_levels) {
                const hrf_level_meta slv = (sg == tseg) ? lv : segs[sg].levels[l];
                const float qc[4] = {q4.x, q4.y, q4.z, q4.w};
                const float2 dy = *(const float2*)(dY_lm + ((size_t)l * n + base + sl) * 2);
                float sv[4][2], w[4];
                uint32_t ci[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float pp = fmaf(qc[v], slv.scale, 0.5f);
                    const float fl = floorf(pp);
                    ci[v] = (uint32_t)(int)fl;
                    w[v] = pp - fl;
                    int c0, c1;
                    float fr;
                    hrf_vec_tap(qc[v], vec_res, c0, c1, fr);
                    const float* vb = vectors + ((size_t)(sg * 4 + v) * vec_res) * ENC_F + 2 * l;
                    const float2 v0 = *(const float2*)(vb + (size_t)c0 * ENC_F), v1 = *(const float2*)(vb + (size_t)c1 * ENC_F);
                    sv[v][0] = v0.x + fr * (v1.x - v0.x);
                    sv[v][1] = v0.y + fr * (v1.y - v0.y);
                }
                if (sg == tseg) {
                    c01 = (ci[0] & 0xFFFFu) | (ci[1] << 16);
                    c23 = (ci[2] & 0xFFFFu) | (ci[3] << 16);
#pragma unroll
                    for (int v = 0; v < 4; ++v) s_w[v][p] = w[v];
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) {
                        const float ga = hrf_through_half(sv[pv[ee]][0] * dy.x * inv_scale, gb, inv_gb);
                        const float gc = hrf_through_half(sv[pv[ee]][1] * dy.y * inv_scale, gb, inv_gb);
                        if constexpr (kHalfG) {     // ga = float(half) * gb exactly: keep the half
                            s_g[ee][0][p] = hrf_boundary_half(sv[pv[ee]][0] * dy.x * inv_scale, inv_gb);
                            s_g[ee][1][p] = hrf_boundary_half(sv[pv[ee]][1] * dy.y * inv_scale, inv_gb);
                        } else {
                            s_g[ee][0][p] = ga;
                            s_g[ee][1][p] = gc;
                        }
                        gm[ee] = fmaxf(gm[ee], fmaxf(fabsf(ga), fabsf(gc)));      // (a NaN is caught by the accumulate kernel's range check)
                    }
                } else {
                    // a sample of another temporal segment than its tile's (the batch was not sorted by segment): its
                    // 4 x 8 corners go to memory directly
                    const hrf_segment_meta* sm = segs + sg;
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) {
                        Corner8 cr;
                        enc_corners(qc[ax[ee][0]], qc[ax[ee][1]], qc[ax[ee][2]], slv, cr);
                        const float g0 = hrf_through_half(sv[pv[ee]][0] * dy.x * inv_scale, gb, inv_gb);
                        const float g1 = hrf_through_half(sv[pv[ee]][1] * dy.y * inv_scale, gb, inv_gb);
                        float* tg = d_tables + 2 * (sm->table_offset + (size_t)ee * sm->entries + slv.offset);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k], cr.w[k] * g0);
                            unsafeAtomicAdd(tg + 2 * (size_t)cr.idx[k] + 1, cr.w[k] * g1);
                        }
                    }
                }
            }
        }
        s_cell[0][p] = c01;
        s_cell[1][p] = c23;
    }
    __syncthreads();
    // The fixed-point unit of the accumulate kernel needs an upper bound of the largest record, not the maximum itself: a
    // record sums w * g over at most SB_RL samples with weights <= 1, so SB_RL x max |g| bounds it (3 bits of the 38 below the
    // largest contribution). Taken here, once per sample, it saves four instructions per RECORD in the walk below, which is
    // bound by vector-ALU issue (profiles/r04_sq_k_scatter_emit_k_scatter_accumulate.txt).
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
        float m = gm[ee];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        if ((tid & 63) == 0 && m > 0.0f) atomicMax(&s_max[ee], __float_as_uint(m * (float)SB_RL * 1.0009765625f));
    }
    const int qshift = sb_queue_shift(lv.size);
    const int sub_shift = 13 - qshift;                       // SB_CT = 2^13 records, split over 2^qshift queues
    if (tile_has_level) {   // (wave-uniform)
        const uint32_t sub_cap = 1u << sub_shift;
        // The walk below is bound by vector-ALU issue (profiles/r04_sq_*): whatever is the same for the whole wavefront is
        // kept scalar. queue(key) = (key >> q_sh) & q_mask covers both chunk maps (contiguous: key >> 13; interleaved: bits
        // [4, 4 + qshift)); the record area of this (tile, level, encoding) is one scalar base, so that a record store is a
        // 32-bit offset (24-bit multiply: slots are below 2^13) instead of a 64-bit address per lane.
        const uint32_t q_sh = qshift <= SB_QCONTIG_LOG2 ? (uint32_t)SB_CHUNK_LOG2 : 4u;
        const uint32_t q_mask = qshift <= SB_QCONTIG_LOG2 ? 0xFFFFFFFFu : ((1u << qshift) - 1u);
        const int e_u = __builtin_amdgcn_readfirstlane(e);          // (e is wave-uniform: SB_RUNS % 64 == 0)
        char* rbase = (char*)(ws.recs + (((size_t)tile * SB_LEVELS + l) * 4 + e_u) * SB_CT);
        float* tg = d_tables + 2 * (segs[tseg].table_offset + (size_t)e_u * segs[tseg].entries + lv.offset);
        uint32_t* cnt = s_cnt[e_u];
        const uint32_t lv_res = lv.res, lv_size = lv.size, lv_hashed = lv.hashed;
        const float* wA = s_w[(e_u == 2) ? 1 : 0];
        const float* wB = s_w[(e_u <= 1) ? 1 : 2];
        const float* wC = s_w[(e_u == 0) ? 2 : 3];
        const GT* g0p = s_g[e_u][0];
        const GT* g1p = s_g[e_u][1];
        // The eight corners of the current cell, each in the slot j = px | py << 1 | pz << 2 of the PARITIES of its
        // coordinates: a cell holds exactly one corner of every parity class, and a corner the next cell shares keeps its
        // coordinates, hence its slot -- nothing moves between registers when the walk changes cell, the slots whose
        // corner changed are emitted and start over.
        float acc[8][2];
        uint32_t key[8];

        // Append the corners selected by `sel` to the queues of their chunks: slots from the LDS counters first (all in
        // flight together), then the 12-byte records; a corner whose queue is full goes to memory directly. (Corners whose
        // sums are exactly zero are queued like the others: the accumulate kernel skips zero addends.)
        auto emit = [&](const bool sel[8]) {
            uint32_t slot[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                slot[j] = 0u;
                if (sel[j]) slot[j] = atomicAdd(&cnt[(key[j] >> q_sh) & q_mask], 1u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (sel[j]) {
                    if (__builtin_expect(slot[j] < sub_cap, 1)) {
                        const uint32_t idx = (((key[j] >> q_sh) & q_mask) << sub_shift) + slot[j];
                        SbRec r; r.key = key[j]; r.a0 = acc[j][0]; r.a1 = acc[j][1];
                        *(SbRec*)(rbase + __umul24(idx, (uint32_t)sizeof(SbRec))) = r;
                    } else {
                        unsafeAtomicAdd(tg + 2 * (size_t)key[j], acc[j][0]);
                        unsafeAtomicAdd(tg + 2 * (size_t)key[j] + 1, acc[j][1]);
                    }
                }
            }
        };
        // cell of a sample on this encoding's three axes (wave-uniform switch) from the packed tile records
        auto cell_of = [&](uint32_t c01, uint32_t c23, uint32_t& ia, uint32_t& ib, uint32_t& ic) {
            if (e_u == 0)      { ia = c01 & 0xFFFFu; ib = c01 >> 16;     ic = c23 & 0xFFFFu; }
            else if (e_u == 1) { ia = c01 & 0xFFFFu; ib = c01 >> 16;     ic = c23 >> 16; }
            else if (e_u == 2) { ia = c01 >> 16;     ib = c23 & 0xFFFFu; ic = c23 >> 16; }
            else               { ia = c01 & 0xFFFFu; ib = c23 & 0xFFFFu; ic = c23 >> 16; }
        };
        // the slots in `fresh` start over on the corners of cell (ia, ib, ic): X[p] / Y[p] / Z[p] = the corner coordinate of
        // parity p on an axis (the cell coordinate itself when its parity is p, else the next one). Entry indices as
        // enc_corners' (tcnn grid_index): mask on hashed levels (their size is a power of two), the stride form on dense
        // ones, which wraps only for far corners of the last cells.
        auto restart = [&](auto hashed_tag, const uint32_t X[2], const uint32_t Y[2], const uint32_t Z[2], const bool fresh[8]) {
            constexpr bool kHashed = decltype(hashed_tag)::value;
            uint32_t hy[2], hz[2];
            if constexpr (kHashed) {
                hy[0] = Y[0] * 2654435761u; hy[1] = Y[1] * 2654435761u;
                hz[0] = Z[0] * 805459861u; hz[1] = Z[1] * 805459861u;
            } else {
                hy[0] = Y[0] * lv_res; hy[1] = Y[1] * lv_res;
                hz[0] = Z[0] * lv_res * lv_res; hz[1] = Z[1] * lv_res * lv_res;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t i;
                if constexpr (kHashed) i = (X[j & 1] ^ hy[(j >> 1) & 1] ^ hz[(j >> 2) & 1]) & (lv_size - 1u);
                else {
                    i = X[j & 1] + hy[(j >> 1) & 1] + hz[(j >> 2) & 1];
                    if (i >= lv_size) { i -= lv_size; if (i >= lv_size) i %= lv_size; }
                }
                key[j] = fresh[j] ? i : key[j];
                acc[j][0] = fresh[j] ? 0.0f : acc[j][0];
                acc[j][1] = fresh[j] ? 0.0f : acc[j][1];
            }
        };
        // corner weights as enc_corners forms them, ((1 * wx) * wy) * wz; the corner of parity p is the cell's low corner
        // on that axis (weight 1 - w) when the cell coordinate has parity p, its high corner (weight w) otherwise
        auto add_sample = [&](int p, uint32_t ia, uint32_t ib, uint32_t ic) {
            const float wa = wA[p], wb = wB[p], wc = wC[p];
            float g0, g1;
            if constexpr (kHalfG) { g0 = __half2float(g0p[p]) * gb; g1 = __half2float(g1p[p]) * gb; }
            else { g0 = g0p[p]; g1 = g1p[p]; }
            const float la = 1.0f - wa, lb = 1.0f - wb, lc = 1.0f - wc;
            const bool oa = ia & 1u, ob = ib & 1u, oc = ic & 1u;
            const float wx[2] = {oa ? wa : la, oa ? la : wa}, wy[2] = {ob ? wb : lb, ob ? lb : wb}, wz[2] = {oc ? wc : lc, oc ? lc : wc};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float w = 1.0f * wx[j & 1];
                w *= wy[(j >> 1) & 1];
                w *= wz[(j >> 2) & 1];
                acc[j][0] = fmaf(w, g0, acc[j][0]);
                acc[j][1] = fmaf(w, g1, acc[j][1]);
            }
        };

        // A run's samples to walk are a prefix of its eight steps (the tile's samples are the first n_here of its 1024) except
        // for samples of another temporal segment, which the staging sent through the direct path: the first step opens the
        // walk (nothing to emit yet), the others compare cells.
        // (one copy of the walk per kind of level: the branch on it is taken once, not once per corner)
        auto walk = [&](auto hashed_tag) {
        uint32_t pa = 0u, pb = 0u, pc = 0u;
        const uint32_t c01_0 = s_cell[0][lane], c23_0 = s_cell[1][lane];
        bool walking = c01_0 != 0xFFFFFFFFu;
        {
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j][0] = 0.0f; acc[j][1] = 0.0f; key[j] = 0u; }
            if (walking) {
                cell_of(c01_0, c23_0, pa, pb, pc);
                uint32_t X[2], Y[2], Z[2];
                bool all[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    X[q] = pa + ((pa ^ (uint32_t)q) & 1u); Y[q] = pb + ((pb ^ (uint32_t)q) & 1u); Z[q] = pc + ((pc ^ (uint32_t)q) & 1u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) all[j] = true;
                restart(hashed_tag, X, Y, Z, all);
                add_sample(lane, pa, pb, pc);
            }
        }
#pragma unroll 1
        for (int k = 1; k < SB_RL; ++k) {
            const int p = k * SB_PAD + lane;
            const uint32_t c01 = s_cell[0][p], c23 = s_cell[1][p];
            if (c01 == 0xFFFFFFFFu) continue;
            uint32_t ia, ib, ic;
            cell_of(c01, c23, ia, ib, ic);
            if (!walking) {
                // (only after a foreign-segment sample at the head of the run: open the walk here)
                walking = true;
                pa = ia; pb = ib; pc = ic;
                uint32_t X[2], Y[2], Z[2];
                bool all[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    X[q] = ia + ((ia ^ (uint32_t)q) & 1u); Y[q] = ib + ((ib ^ (uint32_t)q) & 1u); Z[q] = ic + ((ic ^ (uint32_t)q) & 1u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) all[j] = true;
                restart(hashed_tag, X, Y, Z, all);
            } else if (ia != pa || ib != pb || ic != pc) {
                uint32_t X[2], Y[2], Z[2];
                bool cx[2], cy[2], cz[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    X[q] = ia + ((ia ^ (uint32_t)q) & 1u); cx[q] = X[q] != pa + ((pa ^ (uint32_t)q) & 1u);
                    Y[q] = ib + ((ib ^ (uint32_t)q) & 1u); cy[q] = Y[q] != pb + ((pb ^ (uint32_t)q) & 1u);
                    Z[q] = ic + ((ic ^ (uint32_t)q) & 1u); cz[q] = Z[q] != pc + ((pc ^ (uint32_t)q) & 1u);
                }
                bool gone[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) gone[j] = cx[j & 1] || cy[(j >> 1) & 1] || cz[(j >> 2) & 1];
                emit(gone);
                restart(hashed_tag, X, Y, Z, gone);
                pa = ia; pb = ib; pc = ic;
            }
            add_sample(p, ia, ib, ic);
        }
        if (walking) {
            bool all[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) all[j] = true;
            emit(all);
        }
        };
        if (lv_hashed) walk(std::true_type{}); else walk(std::false_type{});
    }
    __syncthreads();
    if (tid < 4) {
        ws.maxes[((size_t)l * 4 + tid) * ws.tile_cap + tile] = s_max[tid];
        // (bits of a float >= 0 order like the floats; ~90 tiles of a segment raise one word each: nothing next to the record stores)
        if (s_max[tid]) atomicMax(ws.seg_amax + ((size_t)tseg * SB_LEVELS + l) * 4 + tid, s_max[tid]);
    }
    if (tid < 4 * SB_QMAX) {
        const int ee = tid / SB_QMAX, q = tid % SB_QMAX;
        const uint32_t sub_cap = tile_has_level ? (1u << sub_shift) : 0u;
        ws.counts[(((size_t)l * 4 + ee) * SB_QMAX + q) * ws.tile_cap + tile] = min(s_cnt[ee][q], sub_cap);
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate
// ------------------------------------------------------------------------------------------------
#ifndef SB_ACC_THREADS
#define SB_ACC_THREADS 1024
#endif
#ifndef SB_ACC_MINWAVES
#define SB_ACC_MINWAVES 4   // wavefronts per SIMD the register allocation aims at
#endif
#ifndef SB_ACC_UNROLL
#define SB_ACC_UNROLL 8     // tile queues a wavefront reads side by side (records per lane in flight)
#endif
#define SB_SEG_SLOTS 8      // segments the accumulate grid covers at a time
// round-to-nearest-even of a float of magnitude < 2^51 to a 64-bit integer: the float is exact as a double, adding 1.5 * 2^52 rounds it
// to an integer in the double's low mantissa bits (the FPU's own round-to-nearest-even), and the integer is the difference of the bit
// patterns. Same values as __float2ll_rn, which this target expands into a dozen and a half instructions per call -- two calls per
// record, 83 M records per step (accumulate alone on 640 k samples: 0.368 -> 0.362 ms with it, the same sums to the bit: tools/sigbench.py
// prints a checksum. The kernel is not bound by its vector instructions.)
__device__ __forceinline__ long long sb_to_fixed(float x)
{
    const double magic = 6755399441055744.0;                 // 1.5 * 2^52
    return __double_as_longlong((double)x + magic) - __double_as_longlong(magic);
}
// Completion signals of ONE launch over all segments (hrf_scatter_accumulate_signalled; the data-parallel step): the segments are
// dealt to the grid's slots in id order and a slot's workgroups are dispatched before the next slot's, so the table gradients of
// segment 0 are complete long before the launch ends. Every workgroup that is done with a segment of group g -- accumulated it, or
// found nothing queued for it -- adds one to done[g]; a stream that waits for done[g] to reach (segments of g) x (workgroups per
// segment), cumulative over the steps, may read the group's gradients while the same launch is still accumulating the groups behind
// it (hipStreamWaitValue64; tools/microbench/wait_value_probe.hip: the waiter starts ~2 us after the count is reached).
#define SB_SIGNAL_GROUPS 8
struct SbSignal {
    unsigned long long* done;                 // NULL: no signals
    int n;
    int first[SB_SIGNAL_GROUPS], last[SB_SIGNAL_GROUPS];      // group g = segment ids first[g] .. last[g]
};
__global__ __launch_bounds__(SB_ACC_THREADS, SB_ACC_MINWAVES) void k_scatter_accumulate(
    const hrf_segment_meta* __restrict__ segs, int num_segments, SbWorkspace ws, float* __restrict__ d_tables,
    int32_t* __restrict__ flags, int qmax, int n_slots, int seg_first, int seg_count, SbSignal sig)
{
    __shared__ unsigned long long s_acc[2 * SB_CHUNK];     // 128 KB: one workgroup per CU, 16 wavefronts
    // grid: (slot, level, encoding, chunk) with `qmax` = chunks of the model's largest level table (a power of two), levels
    // ascending. (Round 5 measured the two "longest jobs first" orders -- a fine level queues three times the records of a coarse
    // one -- and both lost: finest level first within a slot 0.48-0.50 ms, level-major with the finest first 0.47-0.49 ms, against
    // 0.40-0.44 ms for this order, profiles/r05_scatter_variants.txt. The slot stays the slowest index either way: slots beyond
    // the segments of the batch leave at once, but every such workgroup needs 128 KB of LDS to come free before it can.)
    const int q = (int)(blockIdx.x % (unsigned)qmax);
    const int e = (int)((blockIdx.x / (unsigned)qmax) % 4);
    // (Round 6 tried the finest levels first for the per-group launches of the data-parallel step as well -- a short launch that ends on
    // its longest workgroups -- and it lost again: four group launches 0.83 ms against 0.62 ms in ascending order, profiles/r06_dp_lines.txt.)
    const int l = (int)((blockIdx.x / ((unsigned)qmax * 4)) % SB_LEVELS);
    const int slot = (int)(blockIdx.x / ((unsigned)qmax * 4 * SB_LEVELS));
    (void)n_slots;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = SB_ACC_THREADS / 64;
    // The grid covers SB_SEG_SLOTS segments at a time (a batch holds the frames of at most max_num_frames_per_batch = 8
    // segments; a model of 1 000 frames has 125 and more): workgroup `slot` takes the present segments slot, slot + 8, ...
    // seg_count > 0 (hrf_scatter_accumulate, the data-parallel step): the segments [seg_first, seg_first + seg_count) by id
    // instead of the present ones -- a segment without tiles has nothing queued and its workgroups leave.
    const int n_iter = seg_count > 0 ? seg_count : ws.seg_list[num_segments];
#pragma unroll 1
    for (int si = slot; si < n_iter; si += SB_SEG_SLOTS) {
        const int seg = seg_count > 0 ? seg_first + si : ws.seg_list[si];
        do {                                                        // (one pass; `break` = this workgroup is done with the segment)
        const int t_begin = ws.seg_tile0[seg], t_end = ws.seg_tile0[seg + 1];
        if (t_begin >= t_end) break;                                // (a segment taken by id that owns no tile of this batch)
        if (l >= (int)segs[seg].n_levels) break;
        const hrf_level_meta lv = segs[seg].levels[l];
        const int qshift = sb_queue_shift(lv.size);
        if (q >= (1 << qshift) || sb_entry_of((uint32_t)q, 0u, qshift) >= lv.size) break;         // no entry of the table lies in chunk q
        // fixed-point unit of this (segment, level, encoding): from the largest record any of its tiles queued -- one word the emit
        // kernel raised (round 6; rounds 3-5 scanned the ~90 per-tile maxima here: a global load, an LDS atomic and two barriers in
        // front of every workgroup's first record, in each of the 64 chunk workgroups of the table alike)
        const float amax = __uint_as_float(ws.seg_amax[((size_t)seg * SB_LEVELS + l) * 4 + e]);
        if (!(amax > 0.0f)) break;                              // nothing queued for this table
        // a non-finite record (only after an fp16 overflow upstream, which raises the flag itself): the step must be skipped
        // like GradScaler skips it; nothing is accumulated
        if (!(amax < 3.0e38f)) { if (tid == 0 && flags) atomicOr(flags, 1); break; }
        // amax in [2^ex, 2^(ex+1)); clamped from below so that 2^(SB_FIX_BITS - ex) stays finite (records below 2^-80 --
        // 2^-96 of a gradient before the loss scale -- then keep fewer than 38 bits under the largest one)
        const int ex = max(ilogbf(amax), -80);
        const float to_fix = ldexpf(1.0f, SB_FIX_BITS - ex), from_fix = ldexpf(1.0f, ex - SB_FIX_BITS);
        const uint32_t* cnts = ws.counts + (((size_t)l * 4 + e) * SB_QMAX + q) * ws.tile_cap;
        const int sub_shift = 13 - qshift;
        bool bad = false;
        auto add = [&](const SbRec& r) {
            const uint32_t k = sb_local_of(r.key, qshift);
            bad |= !(fabsf(r.a0) <= amax) || !(fabsf(r.a1) <= amax);      // (a NaN fails the test)
            const long long f0 = sb_to_fixed(r.a0 * to_fix), f1 = sb_to_fixed(r.a1 * to_fix);
            if (f0) atomicAdd(&s_acc[2 * k], (unsigned long long)f0);
            if (f1) atomicAdd(&s_acc[2 * k + 1], (unsigned long long)f1);
        };
        // A wavefront takes the tiles wave, wave + 16, ... of the segment, SB_ACC_UNROLL of them at a time: their queue lengths
        // are wave-uniform (scalar loads, the next group's requested while this group's records are in flight), and one pass
        // reads the same 64-record window of all of them -- up to SB_ACC_UNROLL independent 768-byte reads in flight per
        // wavefront whether the queues are long (8 chunks: ~500 records per tile) or short (64 chunks: ~60). (Round 3 walked
        // one queue at a time; with the short queues of 2^18 / 2^19-entry tables that left a wavefront waiting a memory
        // latency for 60 records: 0.99 ms for the kernel on one 2^18 segment, profiles/r04_*.)
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const SbRec* rb = ws.recs + ((size_t)l * 4 + e) * SB_CT + ((size_t)q << sub_shift);
        constexpr size_t kTileStride = (size_t)SB_LEVELS * 4 * SB_CT;
        int cn[SB_ACC_UNROLL];
        int t0 = t_begin + wave_u;
#pragma unroll
        for (int u = 0; u < SB_ACC_UNROLL; ++u) {
            const int t = t0 + u * kWaves;
            cn[u] = t < t_end ? (int)cnts[t] : 0;
        }
        // (the first queue lengths are on their way while the accumulators are cleared)
        for (int i = tid; i < 2 * SB_CHUNK; i += SB_ACC_THREADS) s_acc[i] = 0ull;
        __syncthreads();
#pragma unroll 1
        for (; t0 < t_end; t0 += SB_ACC_UNROLL * kWaves) {
            int c[SB_ACC_UNROLL], cmax = 0;
#pragma unroll
            for (int u = 0; u < SB_ACC_UNROLL; ++u) { c[u] = cn[u]; cmax = max(cmax, c[u]); }
#pragma unroll
            for (int u = 0; u < SB_ACC_UNROLL; ++u) {
                const int t = t0 + (SB_ACC_UNROLL + u) * kWaves;
                cn[u] = t < t_end ? (int)cnts[t] : 0;
            }
#pragma unroll 1
            for (int w0 = 0; w0 < cmax; w0 += 64) {
                const int i = w0 + lane;
                SbRec r[SB_ACC_UNROLL];
#pragma unroll
                for (int u = 0; u < SB_ACC_UNROLL; ++u) {
                    r[u].key = 0u; r[u].a0 = 0.0f; r[u].a1 = 0.0f;
                    if (i < c[u]) r[u] = rb[(size_t)(t0 + u * kWaves) * kTileStride + i];
                }
#pragma unroll
                for (int u = 0; u < SB_ACC_UNROLL; ++u)
                    if (i < c[u]) add(r[u]);
            }
        }
        if (__any(bad) && lane == 0 && flags) atomicOr(flags, 1);       // a NaN record
        __syncthreads();
        // ONE add per touched entry (unless the direct path wrote to it as well); 16 lanes = one 64-byte request. With the
        // interleaved chunk map a chunk's accumulators are runs of 16 entries (128 bytes of d_tables) 2^qshift runs apart.
        float* tg = d_tables + 2 * (segs[seg].table_offset + (size_t)e * segs[seg].entries + lv.offset);
        for (uint32_t i = tid; i < 2 * SB_CHUNK; i += SB_ACC_THREADS) {
            const long long v = (long long)s_acc[i];
            if (v != 0) {
                const uint32_t entry = sb_entry_of((uint32_t)q, i >> 1, qshift);
                if (entry < lv.size) unsafeAtomicAdd(tg + 2 * (size_t)entry + (i & 1u), (float)v * from_fix);
            }
        }
        __syncthreads();                                                    // (the accumulators are cleared for the next segment)
        } while (0);
        if (sig.done != nullptr) {
            // What the count must be ordered behind is this workgroup's adds into d_tables: agent-scope atomics, performed at the memory
            // side (where every XCD and the copy engines see them), acknowledged to the issuing wavefront before its vmcnt reaches 0 --
            // which every wavefront waits for in front of the barrier. No line of d_tables is dirty in an L2 (nothing is
            // stored, only atomics), so there is nothing for an agent-scope release (an L2 write-back per workgroup) to write back:
            // measured on 640 k samples, one launch 0.39 ms, with this signal 0.39 ms, with one agent-scope release per workgroup
            // 0.49 ms (what four launches cost), with a system-scope fence in every thread 2.86 ms (tools/sigbench.py,
            // SB_SIGNAL_FENCE = 0 / 1 / 2; profiles/r06_signalled_accumulate.txt). The reader starts behind a stream wait and, like every
            // kernel, with an acquire of its own.
#ifndef SB_SIGNAL_FENCE
#define SB_SIGNAL_FENCE 0
#endif
#if SB_SIGNAL_FENCE == 2
            __threadfence();
#endif
            // (The compiler's workgroup-scope release does NOT wait for global memory operations on this target -- the wavefronts of a
            // workgroup share one vector cache -- so the wait is spelled out: every wavefront's outstanding adds are acknowledged
            // before it reaches the barrier.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                int g = -1;
#pragma unroll
                for (int i = 0; i < SB_SIGNAL_GROUPS; ++i)
                    if (i < sig.n && seg >= sig.first[i] && seg <= sig.last[i]) g = i;
                if (g >= 0) {
#if SB_SIGNAL_FENCE == 2
                    __threadfence_system();
#elif SB_SIGNAL_FENCE == 1
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
                    atomicAdd(sig.done + g, 1ull);
                }
            }
        }
    }
}

// chunks of the model's largest level table, rounded up to a power of two: the accumulate grid's innermost extent
static int sb_model_queues(int max_level_entries)
{
    int q = 1;
    while ((int64_t)q * SB_CHUNK < max_level_entries) q <<= 1;
    return q;
}

// The two halves of hrf_encode4d_bwd_tables_binned as calls of their own (ABI 8): the data-parallel step accumulates one group of
// temporal segments at a time and hands each group's table gradients to its reduce-scatter while the next group is still being
// accumulated (trainer.TrainEngine.train_step; SURVEY.md 8(e): "overlap with the remaining backward").
extern "C" int hrf_scatter_emit(const float* xyzt, const int32_t* segment, const float* vectors,
                                const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                const float* d_features_lm, float grad_scale, float grad_boundary, float* d_tables,
                                void* workspace, int64_t workspace_samples, int max_level_entries, hrf_stream_t stream)
{
    if (n == 0) return 0;
    HRF_CHECK_ARG(grad_boundary >= 0.0f, "grad_boundary must be 0 (off) or the factor between the fused and the reference's gradient scale");
    HRF_CHECK_ARG(xyzt && vectors && segments && d_features_lm && d_tables && workspace, "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && num_segments <= SB_MAX_SEGMENTS && vec_res > 1 && grad_scale > 0.0f, "bad arguments (at most 1024 segments)");
    HRF_CHECK_ARG(max_level_entries > 0 && max_level_entries <= SB_QMAX * SB_CHUNK,
                  "level tables above 524288 entries (log2_hashmap_size > 19) are served by hrf_encode4d_bwd (d_features_mode 2)");
    HRF_CHECK_ARG(n <= workspace_samples && n < ((int64_t)1 << 31), "batch larger than the workspace was sized for (hrf_scatter_workspace_bytes)");
    SbWorkspace ws;
    sb_layout(workspace_samples, num_segments, (char*)workspace, &ws);
    const int64_t tiles = sb_tile_cap(n, num_segments);     // upper bound; workgroups beyond the built tiles leave at once
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scatter_tiles, dim3(1), dim3(256), 0, st, segment, n, num_segments, ws);
    if (grad_boundary > 0.0f)
        hipLaunchKernelGGL(k_scatter_emit<true>, dim3((unsigned)(tiles * SB_LEVELS)), dim3(SB_THREADS), 0, st, xyzt, segment, vectors,
                           segments, num_segments, vec_res, n, d_features_lm, 1.0f / grad_scale, d_tables, ws, grad_boundary);
    else
        hipLaunchKernelGGL(k_scatter_emit<false>, dim3((unsigned)(tiles * SB_LEVELS)), dim3(SB_THREADS), 0, st, xyzt, segment, vectors,
                           segments, num_segments, vec_res, n, d_features_lm, 1.0f / grad_scale, d_tables, ws, grad_boundary);
    HRF_CHECK_LAUNCH();
    return 0;
}

// seg_count > 0: the temporal segments [seg_first, seg_first + seg_count) by id; seg_count <= 0: every segment that owns tiles.
extern "C" int hrf_scatter_accumulate(const hrf_segment_meta* segments, int num_segments, float* d_tables, void* workspace,
                                      int64_t workspace_samples, int max_level_entries, int32_t* flags, int seg_first,
                                      int seg_count, hrf_stream_t stream)
{
    HRF_CHECK_ARG(segments && d_tables && workspace, "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && num_segments <= SB_MAX_SEGMENTS, "bad arguments (at most 1024 segments)");
    HRF_CHECK_ARG(max_level_entries > 0 && max_level_entries <= SB_QMAX * SB_CHUNK, "level tables above 524288 entries");
    HRF_CHECK_ARG(seg_count <= 0 || (seg_first >= 0 && seg_first + seg_count <= num_segments), "segment range outside the model");
    SbWorkspace ws;
    sb_layout(workspace_samples, num_segments, (char*)workspace, &ws);
    const int span = seg_count > 0 ? seg_count : num_segments;
    const int slots = span < SB_SEG_SLOTS ? span : SB_SEG_SLOTS;
    const int qmax = sb_model_queues(max_level_entries);
    SbSignal none;
    none.done = nullptr; none.n = 0;
    hipLaunchKernelGGL(k_scatter_accumulate, dim3((unsigned)(slots * SB_LEVELS * 4 * qmax)), dim3(SB_ACC_THREADS), 0,
                       (hipStream_t)stream, segments, num_segments, ws, d_tables, flags, qmax, slots, seg_first,
                       seg_count > 0 ? seg_count : 0, none);
    HRF_CHECK_LAUNCH();
    return 0;
}

// Workgroups that report on one segment in hrf_scatter_accumulate_signalled: the count a waiter multiplies by the segments of a group.
extern "C" int64_t hrf_scatter_signals_per_segment(int max_level_entries)
{
    if (max_level_entries <= 0 || max_level_entries > SB_QMAX * SB_CHUNK) return -1;
    return (int64_t)SB_LEVELS * 4 * sb_model_queues(max_level_entries);
}

// ONE accumulate launch over every temporal segment of the model by id, with completion signals per GROUP of consecutive segment
// ids (group_bounds: n_groups pairs (first id, last id) in HOST memory, ascending and disjoint; at most 8): group_done[g] (device,
// 64-bit, never reset by this library) grows by hrf_scatter_signals_per_segment() for each segment of group g, as soon as that
// segment's gradients are complete in d_tables.
extern "C" int hrf_scatter_accumulate_signalled(const hrf_segment_meta* segments, int num_segments, float* d_tables, void* workspace,
                                                int64_t workspace_samples, int max_level_entries, int32_t* flags,
                                                const int32_t* group_bounds, int n_groups, uint64_t* group_done,
                                                hrf_stream_t stream)
{
    HRF_CHECK_ARG(segments && d_tables && workspace && group_bounds && group_done, "NULL argument");
    HRF_CHECK_ARG(num_segments > 0 && num_segments <= SB_MAX_SEGMENTS, "bad arguments (at most 1024 segments)");
    HRF_CHECK_ARG(max_level_entries > 0 && max_level_entries <= SB_QMAX * SB_CHUNK, "level tables above 524288 entries");
    HRF_CHECK_ARG(n_groups >= 1 && n_groups <= SB_SIGNAL_GROUPS, "1..8 signal groups");
    SbSignal sig;
    sig.done = (unsigned long long*)group_done; sig.n = n_groups;
    int prev = -1;
    for (int g = 0; g < SB_SIGNAL_GROUPS; ++g) {
        sig.first[g] = g < n_groups ? group_bounds[2 * g] : 0;
        sig.last[g] = g < n_groups ? group_bounds[2 * g + 1] : -1;
        if (g < n_groups) {
            HRF_CHECK_ARG(sig.first[g] > prev && sig.last[g] >= sig.first[g] && sig.last[g] < num_segments,
                          "signal groups must be ascending, disjoint ranges of segment ids");
            prev = sig.last[g];
        }
    }
    SbWorkspace ws;
    sb_layout(workspace_samples, num_segments, (char*)workspace, &ws);
    const int slots = num_segments < SB_SEG_SLOTS ? num_segments : SB_SEG_SLOTS;
    const int qmax = sb_model_queues(max_level_entries);
    hipLaunchKernelGGL(k_scatter_accumulate, dim3((unsigned)(slots * SB_LEVELS * 4 * qmax)), dim3(SB_ACC_THREADS), 0,
                       (hipStream_t)stream, segments, num_segments, ws, d_tables, flags, qmax, slots, 0, num_segments, sig);
    HRF_CHECK_LAUNCH();
    return 0;
}

// A stream waits until the 64-bit value at `addr` (device memory) is >= value: hipStreamWaitValue64. hrf_can_stream_wait_value():
// 1 when the device supports it.
extern "C" int hrf_can_stream_wait_value(void)
{
    int dev = 0, can = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) return 0;
    return can ? 1 : 0;
}
extern "C" int hrf_stream_wait_value64(hrf_stream_t stream, const uint64_t* addr, uint64_t value)
{
    HRF_CHECK_ARG(addr, "NULL address");
    if (hipStreamWaitValue64((hipStream_t)stream, (void*)addr, value, hipStreamWaitValueGte, ~0ull) != hipSuccess) {
        hrf_set_error("%s: hipStreamWaitValue64 failed", __func__);
        return 2;
    }
    return 0;
}

extern "C" int hrf_encode4d_bwd_tables_binned(const float* xyzt, const int32_t* segment, const float* vectors,
                                              const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                              const float* d_features_lm, float grad_scale, float grad_boundary,
                                              float* d_tables, void* workspace, int64_t workspace_samples,
                                              int max_level_entries, int32_t* flags, hrf_stream_t stream)
{
    if (n == 0) return 0;
    if (int rc = hrf_scatter_emit(xyzt, segment, vectors, segments, num_segments, vec_res, n, d_features_lm, grad_scale,
                                  grad_boundary, d_tables, workspace, workspace_samples, max_level_entries, stream))
        return rc;
    return hrf_scatter_accumulate(segments, num_segments, d_tables, workspace, workspace_samples, max_level_entries, flags, 0, 0,
                                  stream);
}
