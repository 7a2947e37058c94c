// Device helpers shared by the hash-encoding kernels (encode.hip, march.hip): tcnn HashGrid indexing for the
// four encodings of Decomposition4D (decomposition4d.py:126-129).
#pragma once
#include "hrf_common.h"

#define ENC_TILE 64
#define ENC_F 32  // features per sample (16 levels x 2)

// acc += w * half(lo / hi 16 bits of `pair`). Default: two v_cvt_f32_f16 + one v_pk_fma_f32 per table entry (what the compiler
// makes of the plain expression). -DENC_FMA_MIX: v_fma_mix_f32 converts the half operand inside the FMA (same value, no
// conversions, two instructions instead of three) -- measured on MI355X (round 4, profiles/r04_ab_fma_mix_gather.txt): k_prune_march 1.028 ms
// against 0.958 ms, k_encode4d_fwd unchanged; the mixed-precision FMA does not issue at the rate of the packed one. Not used.
__device__ __forceinline__ void enc_fma_half2(float w, uint32_t pair, float& f0, float& f1)
{
#ifdef ENC_FMA_MIX
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(f0) : "v"(w), "v"(pair));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(f1) : "v"(w), "v"(pair));
#else
    const float2 vf = __half22float2(__builtin_bit_cast(__half2, pair));
    f0 = fmaf(w, vf.x, f0);
    f1 = fmaf(w, vf.y, f1);
#endif
}


// Order in which a wavefront walks its N level groups (march: 8 pairs of levels; k_encode4d_fwd: 4 groups of four levels spread over
// the workgroup's wavefronts). Levels are independent and each writes its own columns of the LDS feature row, so any order gives the
// same bits. ENC_PHASE = 1 (shipped since round 6): the ascending walk starts at the group the 100 MHz chip clock points at,
// (clock >> ENC_PHASE_SHIFT) & (N - 1) -- wavefronts that start a step within the same 20 us therefore gather from the same one or
// two levels' tables (1-4 MB of a segment's 8-17 MB) and an XCD's 4 MB L2 keeps them: the march's L2 misses fall by 36-40 %, its
// time by 2-3 %, the render-pass encode's by 8-10 % (profiles/r06_l2_phase_go_nogo.txt: the no-miss bound of these kernels is
// only 13-14 % below where they run, which is why the elaborate form -- a per-XCD phase barrier -- was not built).
// 0 = ascending (rounds 1-5). Measurement builds: 2 = every iteration takes the pending group nearest to the current phase,
// 3 = as 2, but a wavefront that is ahead of the clock sleeps until its group comes up (loses at every shift);
// -DENC_PHASE_TUNE reads the shift from HRF_PHASE_SHIFT / HRF_PHASE_SHIFT_FWD at launch.
#ifndef ENC_PHASE
#define ENC_PHASE 1
#endif
#ifndef ENC_PHASE_SHIFT
#define ENC_PHASE_SHIFT 11     // 2^11 ticks of 10 ns per phase
#endif
template <int N>
__device__ __forceinline__ int enc_phase_next(int k, int shift, uint32_t& done)
{
#if ENC_PHASE == 0
    return k;
#else
    constexpr uint32_t M = N - 1;
    uint32_t ph = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> shift) & M;
#if ENC_PHASE == 1
    if (k == 0) done = ph;          // (the rotation of this walk)
    return (int)((done + (uint32_t)k) & M);
#else
    constexpr uint32_t ALL = (1u << N) - 1u;
    const uint32_t pend = ~done & ALL;
#if ENC_PHASE == 3
    for (int spin = 0; spin < 64 && !((pend >> ph) & 1u); ++spin) {
        __builtin_amdgcn_s_sleep(32);
        ph = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> shift) & M;
    }
#endif
    const uint32_t rot = ((pend >> ph) | (pend << (N - ph))) & ALL;
    const int g = (int)((ph + (uint32_t)__builtin_ctz(rot)) & M);
    done |= 1u << g;
    return g;
#endif
#endif
}

struct EncCoords {
    float c[4];  // x, y, z, t in [0,1]
};

// coordinates of encoding e: 0 xyz, 1 xyt, 2 yzt, 3 xzt  (decomposition4d.py:126-129)
__device__ __forceinline__ void enc_pick(const EncCoords& q, int e, float& a, float& b, float& c)
{
    switch (e) {
        case 0: a = q.c[0]; b = q.c[1]; c = q.c[2]; break;
        case 1: a = q.c[0]; b = q.c[1]; c = q.c[3]; break;
        case 2: a = q.c[1]; b = q.c[2]; c = q.c[3]; break;
        default: a = q.c[0]; b = q.c[2]; c = q.c[3]; break;
    }
}

struct Corner8 {
    uint32_t idx[8];
    float w[8];
};

// tcnn pos_fract + grid_index for the 8 corners of one (encoding, level)  (A.1)
__device__ __forceinline__ void enc_corners(float a, float b, float c, const hrf_level_meta& lv, Corner8& out)
{
    const float pa = fmaf(a, lv.scale, 0.5f), pb = fmaf(b, lv.scale, 0.5f), pc = fmaf(c, lv.scale, 0.5f);
    const float fa = floorf(pa), fb = floorf(pb), fc = floorf(pc);
    const float wa = pa - fa, wb = pb - fb, wc = pc - fc;
    const uint32_t ia = (uint32_t)(int)fa, ib = (uint32_t)(int)fb, ic = (uint32_t)(int)fc;
    const uint32_t size = lv.size, res = lv.res;
    if (lv.hashed) {
        const uint32_t mask = size - 1;  // hashed levels have size == 2^log2_hashmap_size (checked on the host)
        const uint32_t hb0 = ib * 2654435761u, hb1 = (ib + 1) * 2654435761u;
        const uint32_t hc0 = ic * 805459861u, hc1 = (ic + 1) * 805459861u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = ia + (k & 1);
            const uint32_t hy = (k & 2) ? hb1 : hb0;
            const uint32_t hz = (k & 4) ? hc1 : hc0;
            out.idx[k] = (x ^ hy ^ hz) & mask;
        }
    } else {
        const uint32_t rr = res * res;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t i = (ia + (k & 1)) + (ib + ((k >> 1) & 1)) * res + (ic + ((k >> 2) & 1)) * rr;
            if (i >= size) { i -= size; if (i >= size) i %= size; }
            out.idx[k] = i;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float w = 1.0f;
        w *= (k & 1) ? wa : (1.0f - wa);
        w *= (k & 2) ? wb : (1.0f - wb);
        w *= (k & 4) ? wc : (1.0f - wc);
        out.w[k] = w;
    }
}


// fp32 accumulate, THEN one rounding to half: the definition the oracle states (hashgrid_encode, accumulate="fp32"). Left alone,
// the compiler folds the last fused multiply-add and the conversion into v_fma_mixlo_f16 in some instantiations and not in others,
// and that instruction rounds the exact a*b+c to half ONCE -- a different value in ~1 of 20 000 cases than the fp32-rounded sum
// rounded again (round 5: k_encode4d_fwd<false> and the march had it, k_encode4d_fwd<true> and k_hashgrid_fwd did not, so the prune
// pass and the render pass of one step saw features one half ulp apart now and then). The empty asm pins the fp32 sums in
// registers: every instantiation converts the same fp32 values.
__device__ __forceinline__ void enc_pin_f32(float& a, float& b)
{
    asm("" : "+v"(a), "+v"(b));      // (not volatile: pinned values, free scheduling)
}

// Table entry through a wavefront-uniform base + a 32-bit byte offset (one address VGPR per load instead of two;
// an encoding's table is far below 4 GB).
__device__ __forceinline__ __half2 enc_entry(const __half2* __restrict__ tb, uint32_t idx)
{
    return *(const __half2*)((const char*)tb + (idx << 2));
}

// Trilinear feature pair of one (level, encoding) -- tcnn kernel_grid forward (A.1): eight independent 4-byte
// gathers, fp32 fmaf accumulation over the corners in tcnn's order, result NOT yet rounded to half.
// (Fetching x-neighbour corners with one 8-byte load when their entries are adjacent -- always on dense levels, for
// even x on hashed ones -- was measured twice on MI355X: the march drops from 0.58 to 0.44 of the byte roofline.
// A divergent 8-byte gather costs the texture-address path more than two 4-byte ones save.)
__device__ __forceinline__ void enc_gather(const __half2* __restrict__ tb, float a, float b, float c,
                                           const hrf_level_meta& lv, float& f0, float& f1)
{
    Corner8 cr;
    enc_corners(a, b, c, lv, cr);
    __half2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = enc_entry(tb, cr.idx[k]);
    f0 = 0.0f; f1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) enc_fma_half2(cr.w[k], __builtin_bit_cast(uint32_t, v[k]), f0, f1);
    enc_pin_f32(f0, f1);     // (callers round these fp32 sums to half; see enc_pin_f32)
}

// Cooperative variant for lanes that hold CONSECUTIVE samples of a ray (march steps): on the coarse and middle levels
// many neighbouring lanes fall into the same grid cell (64 march samples span 0.026 units: 2-3 cells at level 0, ~20 at
// level 10), and a gather is charged per active lane (masked-off lanes are free, profiles/r01_microbench_load_width_lanes.txt).
// Only the first lane of each run of equal cells ("head") fetches the eight corners; the others take the head's
// values through the LDS crossbar (ds_bpermute, no LDS memory). Weights stay per lane, so the result is bit-identical
// to enc_gather. All four encodings of the level are issued before any value is consumed (one gather latency per
// level, as in the plain path). `le_mask` = bits [0, lane] set. The cell key keeps 10 bits per axis: keys are only
// compared between ADJACENT lanes; when those are consecutive samples of ONE ray (the march) their cells are a few
// cells apart at most and the truncation cannot alias. Callers whose adjacent lanes may belong to different rays
// (k_encode4d_fwd: consecutive samples of the batch) pass wide_key = true on levels with more than 1024 cells per
// axis: the bits above the tenth are then compared as well (wavefront-uniform branch, one more DPP move).
// Measured (march, MI355X): 0.62 -> 0.74 of the byte roofline once the rays are scheduled by frame over the XCDs
// (before that the kernel sat on the fabric line rate and this changed nothing).
// A register whose content does not matter (lanes that never read it): no instruction is emitted for it. (Initialising the
// gathered values of the non-head lanes below cost 32 v_mov per level, 8 % of the level body's vector instructions.)
__device__ __forceinline__ uint32_t enc_any_u32()
{
    uint32_t v;
    asm volatile("" : "=v"(v));
    return v;
}

typedef float enc_f2 __attribute__((ext_vector_type(2)));


// Round 5: the level body is bound by vector-ALU issue (profiles/r04_sq_k_prune_march.txt), so it is written around its
// instruction count -- same values, bit for bit, as enc_corners + enc_gather per encoding:
//   * corner positions are formed as BYTE offsets: (x ^ y P1 ^ z P2) & mask, shifted left by two, is
//     ((4 x) ^ (y 4 P1) ^ (z 4 P2)) & (4 mask) -- the shift distributes over xor / and, the products wrap modulo 2^32 either
//     way -- and the stride form of the dense levels is linear; 32 shifts per level gone;
//   * the gathered values of lanes that are not the head of their run are never read (ds_bpermute pulls from head lanes
//     only): they are left undefined instead of zeroed;
//   * the corner weights ((1 wx) wy) wz are formed as packed pairs over the x corner (v_pk_mul_f32): 6 instructions per
//     encoding instead of 12-16, the (x, y) products shared by the xyz and xyt encodings.
__device__ __forceinline__ void enc_level_shared(const EncCoords& q, const __half2* __restrict__ tbase, uint32_t entries,
                                                 const hrf_level_meta& lv, unsigned long long le_mask, float fe[4][2],
                                                 int table_key = 0, bool wide_key = false)
{
    // table_key: anything besides the cell that selects the table (the segment, when lanes may differ in it)
    const bool new_table = table_key != __builtin_amdgcn_update_dpp(-1, table_key, 0x138, 0xf, 0xf, false);
    // per axis: cell, fraction (tcnn pos_fract, as enc_corners)
    uint32_t ci[4];
    float wf[4], lf[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float pp = fmaf(q.c[v], lv.scale, 0.5f);
        const float fl = floorf(pp);
        ci[v] = (uint32_t)(int)fl;
        wf[v] = pp - fl;
        lf[v] = 1.0f - wf[v];
    }
    const int ax[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};   // decomposition4d.py:126-129
    const uint32_t size = lv.size, res = lv.res;
    uint32_t v[4][8];
    int head_lane[4];
    enc_f2 wk[4][4];        // corner weights of encoding e: wk[e][k >> 1] = (w[k], w[k + 1]), k even
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int A = ax[e][0], B = ax[e][1], C = ax[e][2];
        const uint32_t ia = ci[A], ib = ci[B], ic = ci[C];
        uint32_t off[8];    // byte offsets of the eight corners inside the encoding's level table
        if (lv.hashed) {
            const uint32_t mask4 = (size - 1u) << 2;  // hashed levels have size == 2^log2_hashmap_size (checked on the host)
            const uint32_t x0 = ia << 2, x1 = x0 + 4u;
            const uint32_t hb0 = ib * (2654435761u << 2), hb1 = hb0 + (2654435761u << 2);
            const uint32_t hc0 = ic * (805459861u << 2), hc1 = hc0 + (805459861u << 2);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                off[k] = (((k & 1) ? x1 : x0) ^ ((k & 2) ? hb1 : hb0) ^ ((k & 4) ? hc1 : hc0)) & mask4;
        } else {
            const uint32_t size4 = size << 2, r4 = res << 2, rr4 = res * res << 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t i = ((ia + (k & 1)) << 2) + (ib + ((k >> 1) & 1)) * r4 + (ic + ((k >> 2) & 1)) * rr4;
                if (i >= size4) { i -= size4; if (i >= size4) i %= size4; }
                off[k] = i;
            }
        }
#ifdef ENC_FOLD_MASK      // measurement only (WRONG values): every gather lands in a window that stays in L2 -- the no-miss bound
#pragma unroll
        for (int k = 0; k < 8; ++k) off[k] &= (uint32_t)(ENC_FOLD_MASK);
#endif
        // run heads: keys are only compared between adjacent lanes. Without wide_key the three cells are packed by two
        // shift-adds (10 bits apart): callers pass wide_key = false only when the cells are below 1024 (then this is the
        // masked key) or when adjacent lanes are consecutive samples of one ray (cells a few apart: the sum differs whenever a
        // cell does). wide_key: 10 bits per axis masked + the bits above compared as well.
        int key;
        bool head;
        if (wide_key) {
            key = (int)((ia & 1023u) | ((ib & 1023u) << 10) | ((ic & 1023u) << 20));
            const int hi = (int)((ia >> 10) | ((ib >> 10) << 10) | ((ic >> 10) << 20));
            head = (key != __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false)) || new_table ||
                   (hi != __builtin_amdgcn_update_dpp(-1, hi, 0x138, 0xf, 0xf, false));
        } else {
            key = (int)(ia + (ib << 10) + (ic << 20));
            // previous lane's key (wave_shr:1; lane 0 keeps ~key, which differs from key)
            head = (key != __builtin_amdgcn_update_dpp(~key, key, 0x138, 0xf, 0xf, false)) || new_table;
        }
        const unsigned long long H = __ballot(head);
        head_lane[e] = (63 - __builtin_clzll(H & le_mask)) << 2;
        const char* tb = (const char*)(tbase + (size_t)e * entries + lv.offset);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[e][k] = enc_any_u32();
        if (head) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[e][k] = *(const uint32_t*)(tb + off[k]);
        }
        // weights in enc_corners' order: ((1 * wx) * wy) * wz
        const enc_f2 xa = {lf[A], wf[A]};
        const enc_f2 ab0 = xa * lf[B], ab1 = xa * wf[B];
        wk[e][0] = ab0 * lf[C]; wk[e][1] = ab1 * lf[C];
        wk[e][2] = ab0 * wf[C]; wk[e][3] = ab1 * wf[C];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t sv = (uint32_t)__builtin_amdgcn_ds_bpermute(head_lane[e], (int)v[e][k]);
            enc_fma_half2((k & 1) ? wk[e][k >> 1].y : wk[e][k >> 1].x, sv, f0, f1);
        }
        enc_pin_f32(f0, f1);
        const float2 hf = __half22float2(__floats2half2_rn(f0, f1));
        fe[e][0] = hf.x; fe[e][1] = hf.y;
    }
}

// The level body with PAIRED x-neighbour fetches (round 6, -DENC_PAIR=1; a measurement build, NOT shipped: profiles/r06_pair_fetch_no_go.txt).
// A gather costs a CU by the distinct cache lines it looks up (profiles/r01_microbench_load_width_lanes.txt), and on hashed levels
// tcnn's index is x ^ (y P1) ^ (z P2): for even x0 the x-neighbour corners (x0, y, z), (x0 + 1, y, z) are the halves of ONE aligned
// 8-byte pair. Head lanes fetch, per (y, z) corner pair, the aligned 8 bytes that hold corner x0, and only the lanes with odd x0
// issue a 4-byte gather for the second corner: 1.5 line look-ups per pair instead of 2. In isolation that pays as the line count
// says (profiles/r06_microbench_pair_loads.txt: -24 % per corner pair at 16-64 active lanes); in the kernels it loses -- march
// 0.923 -> 1.122 ms, 1.063 ms when only (encoding, level)s with >= 24 head lanes use pairs (-DENC_PAIR_MIN=24): the instruction
// count is unchanged, the halves have to be sorted into corner order (+9 % vector instructions), 16 more registers are in flight,
// and the line look-ups it saves are not what the march waits for. Dense levels keep 4-byte gathers (gfx950 aligns the address of
// a global_load_dwordx2 down to 8 bytes: an entry pair at an odd index cannot be fetched as one). Head lanes sort the halves into
// corner order before the ds_bpermute hand-out, so values, weights and fmaf order are those of enc_level_shared: bit-identical
// features (the parity suite passes on the variant library).
#ifndef ENC_PAIR_MIN
#define ENC_PAIR_MIN 0     // head lanes an (encoding, level) needs before its corners are fetched as pairs (below: two 4-byte gathers)
#endif
__device__ __forceinline__ void enc_level_shared_pair(const EncCoords& q, const __half2* __restrict__ tbase, uint32_t entries,
                                                      const hrf_level_meta& lv, unsigned long long le_mask, float fe[4][2],
                                                      int table_key = 0, bool wide_key = false)
{
    const bool new_table = table_key != __builtin_amdgcn_update_dpp(-1, table_key, 0x138, 0xf, 0xf, false);
    uint32_t ci[4];
    float wf[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float pp = fmaf(q.c[v], lv.scale, 0.5f);
        const float fl = floorf(pp);
        ci[v] = (uint32_t)(int)fl;
        wf[v] = pp - fl;
    }
    const int ax[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};   // decomposition4d.py:126-129
    const uint32_t size = lv.size, res = lv.res;
    uint2 va[4][4];          // [encoding][y, z corner pair]: paired: the aligned 8 bytes that hold corner x0; plain: (corner x0, corner x0 + 1)
    uint32_t vb[4][4];       // paired: corner x0 + 1 of the lanes whose x0 is odd (it lies in another pair)
    int head_lane[4];
    bool paired[4];          // (wave-uniform) this encoding's corners were fetched as pairs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int A = ax[e][0], B = ax[e][1], C = ax[e][2];
        const uint32_t ia = ci[A], ib = ci[B], ic = ci[C];
        uint32_t o0[4], o1[4];   // byte offsets of corners (x0, y, z), (x0 + 1, y, z) for the four (y, z)
        if (lv.hashed) {
            const uint32_t mask4 = (size - 1u) << 2;
            const uint32_t x0 = ia << 2, x1 = x0 + 4u;
            const uint32_t hb0 = ib * (2654435761u << 2), hb1 = hb0 + (2654435761u << 2);
            const uint32_t hc0 = ic * (805459861u << 2), hc1 = hc0 + (805459861u << 2);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t h = ((p & 1) ? hb1 : hb0) ^ ((p & 2) ? hc1 : hc0);
                o0[p] = (x0 ^ h) & mask4;
                o1[p] = (x1 ^ h) & mask4;
            }
        } else {
            const uint32_t size4 = size << 2, r4 = res << 2, rr4 = res * res << 2;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t i = (ia << 2) + (ib + (uint32_t)(p & 1)) * r4 + (ic + (uint32_t)((p >> 1) & 1)) * rr4, j = i + 4u;
                if (i >= size4) { i -= size4; if (i >= size4) i %= size4; }
                if (j >= size4) { j -= size4; if (j >= size4) j %= size4; }
                o0[p] = i; o1[p] = j;
            }
        }
        int key;
        bool head;
        if (wide_key) {
            key = (int)((ia & 1023u) | ((ib & 1023u) << 10) | ((ic & 1023u) << 20));
            const int hi = (int)((ia >> 10) | ((ib >> 10) << 10) | ((ic >> 10) << 20));
            head = (key != __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false)) || new_table ||
                   (hi != __builtin_amdgcn_update_dpp(-1, hi, 0x138, 0xf, 0xf, false));
        } else {
            key = (int)(ia + (ib << 10) + (ic << 20));
            head = (key != __builtin_amdgcn_update_dpp(~key, key, 0x138, 0xf, 0xf, false)) || new_table;
        }
        const unsigned long long H = __ballot(head);
        head_lane[e] = (63 - __builtin_clzll(H & le_mask)) << 2;
        // pairs only on hashed levels (a dense index is 4-byte aligned at best) and only where enough lanes gather: below ~8 active
        // lanes an 8-byte gather costs more than the 4-byte one it replaces (profiles/r06_microbench_pair_loads.txt)
        paired[e] = lv.hashed && __popcll(H) >= ENC_PAIR_MIN;
        const char* tb = (const char*)(tbase + (size_t)e * entries + lv.offset);
#pragma unroll
        for (int p = 0; p < 4; ++p) { va[e][p].x = enc_any_u32(); va[e][p].y = enc_any_u32(); vb[e][p] = enc_any_u32(); }
        if (paired[e]) {
            if (head) {
#pragma unroll
                for (int p = 0; p < 4; ++p) va[e][p] = *(const uint2*)(tb + (o0[p] & ~4u));   // holds corner x0 + 1 as well when x0 is even
                if (ia & 1u) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) vb[e][p] = *(const uint32_t*)(tb + o1[p]);
                }
            }
        } else if (head) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                va[e][p].x = *(const uint32_t*)(tb + o0[p]);
                va[e][p].y = *(const uint32_t*)(tb + o1[p]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int A = ax[e][0], B = ax[e][1], C = ax[e][2];
        // which half of the aligned 8 bytes is corner x0: bit 0 of its index x0 ^ (y P1) ^ (z P2) -- P1, P2 are odd, so the parity of
        // x0 + y + z; its x neighbour is the other half when x0 is even and sits in vb when x0 is odd
        const uint32_t par = (ci[A] ^ ci[B] ^ ci[C]) & 1u;
        const bool odd = (ci[A] & 1u) != 0u;
        // weights in enc_corners' order: ((1 * wx) * wy) * wz
        const float la = 1.0f - wf[A], lb = 1.0f - wf[B], lc = 1.0f - wf[C];
        const enc_f2 xa = {la, wf[A]};
        const enc_f2 ab0 = xa * lb, ab1 = xa * wf[B];
        const enc_f2 wk[4] = {ab0 * lc, ab1 * lc, ab0 * wf[C], ab1 * wf[C]};
        float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t c0 = va[e][p].x, c1 = va[e][p].y;
            if (paired[e]) {     // (wave-uniform)
                const bool hi_first = ((par ^ (uint32_t)(p & 1) ^ (uint32_t)((p >> 1) & 1)) & 1u) != 0u;
                c0 = hi_first ? va[e][p].y : va[e][p].x;
                c1 = odd ? vb[e][p] : (hi_first ? va[e][p].x : va[e][p].y);
            }
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_ds_bpermute(head_lane[e], (int)c0);
            const uint32_t s1 = (uint32_t)__builtin_amdgcn_ds_bpermute(head_lane[e], (int)c1);
            enc_fma_half2(wk[p].x, s0, f0, f1);
            enc_fma_half2(wk[p].y, s1, f0, f1);
        }
        enc_pin_f32(f0, f1);
        const float2 hf = __half22float2(__floats2half2_rn(f0, f1));
        fe[e][0] = hf.x; fe[e][1] = hf.y;
    }
}

#ifndef ENC_PAIR
#define ENC_PAIR 0
#endif
#if ENC_PAIR
#define ENC_LEVEL_SHARED enc_level_shared_pair
#else
#define ENC_LEVEL_SHARED enc_level_shared
#endif

// The four encodings of one level WITHOUT cell sharing: 32 independent gathers per lane, all issued before any is consumed.
// Same values as enc_level_shared (and as enc_gather per encoding). On the finest levels consecutive march samples hardly ever
// share a cell (step 4e-4 x res 2048 = 0.8 cells per sample and axis), so the head-lane bookkeeping of the shared form -- key,
// DPP compare, ballot, head lane, 8 ds_bpermute per encoding: ~25 of a level's ~75 vector instructions per encoding -- buys no
// saved gather there. (Round 4 argued from there that the plain form must win on fine levels; it does not, 0.950 -> 0.966 ms for the
// march, and round 5 found out why: the kernels are bound by their gathers' cache-line requests, not by instructions,
// profiles/r05_gather_bound_ablations.txt. Kept for the -DMARCH_PLAIN_FROM_LEVEL / -DFWD_PLAIN_FROM_LEVEL comparison builds.)
__device__ __forceinline__ void enc_level_plain(const EncCoords& q, const __half2* __restrict__ tbase, uint32_t entries,
                                                const hrf_level_meta& lv, float fe[4][2])
{
    uint32_t v[4][8];
    Corner8 cr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a, b, c;
        enc_pick(q, e, a, b, c);
        enc_corners(a, b, c, lv, cr[e]);
        const __half2* tb = tbase + (size_t)e * entries + lv.offset;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[e][k] = __builtin_bit_cast(uint32_t, enc_entry(tb, cr[e].idx[k]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) enc_fma_half2(cr[e].w[k], v[e][k], f0, f1);
        enc_pin_f32(f0, f1);
        const float2 hf = __half22float2(__floats2half2_rn(f0, f1));
        fe[e][0] = hf.x; fe[e][1] = hf.y;
    }
}

// Tried and measured on MI355X, not kept (see DESIGN.md "What did not pay"): sharing the corner fetches through LDS
// (run detection + 8*runs-lane gather + LDS broadcast): a dependent shuffle -> LDS -> gather -> LDS chain per
// (level, encoding), 20 % slower than plain gathers.
