// What does the record emission of k_scatter_emit (csrc/scatter.hip) cost by itself, and what would the alternatives cost?
// The emit kernel is NOT bound by one thing (round 4, profiles/r04_sq_k_scatter_emit_k_scatter_accumulate_rewritten.txt: VALU issue 62 %,
// 51 M L2 write requests per launch from 4.9 M store instructions, 8-way conflicts on its LDS slot counters), and taking 14 % of its
// vector instructions away left its time where it was. This micro-benchmark isolates the store side: every workgroup plays one
// (tile, level) of the emit kernel -- 512 threads, four "encodings" of 128 threads, each thread appends RECS records to one of Q
// queues of its encoding (queue drawn per record from a hash, as the chunk of a hashed table entry is), slots from LDS counters.
//   direct12   what the kernel does: 12-byte records, global_store_dwordx3 at (queue << sub_shift) + slot
//   direct16   the same with 16-byte records (one aligned dwordx4 per record: never straddles a 64-byte line)
//   staged12   records go to an LDS ring per (wavefront, queue) first (RING slots); a ring is written out as one contiguous burst
//              by its wavefront when it fills, and at the end
// Prints ns per record and records per second for Q = 8 / 32 / 64. Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/record_store_bench.hip -o /tmp/record_store_bench && /tmp/record_store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CT 8192          // record capacity per (workgroup, encoding), as SB_CT
#define RECS 32          // records per thread: 128 threads x 32 = 4096 per encoding, the finest level's load
#define RING 16          // staged: records per (wavefront, queue) ring

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int BYTES>
__global__ __launch_bounds__(512) void k_direct(char* out, uint32_t* counts, int qshift)
{
    __shared__ uint32_t s_cnt[4][64];
    const int tid = threadIdx.x, e = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (tid < 256) s_cnt[tid >> 6][tid & 63] = 0;
    __syncthreads();
    const int sub_shift = 13 - qshift;
    char* base = out + ((size_t)blockIdx.x * 4 + e) * CT * BYTES;
    uint32_t h = mix(blockIdx.x * 512u + tid);
#pragma unroll 4
    for (int i = 0; i < RECS; ++i) {
        h = mix(h + i);
        const uint32_t q = h & ((1u << qshift) - 1u);
        const uint32_t slot = atomicAdd(&s_cnt[e][q], 1u);
        if (slot < (1u << sub_shift)) {
            const uint32_t idx = (q << sub_shift) + slot;
            if (BYTES == 12) { struct R { uint32_t k; float a, b; } r = {h, (float)i, (float)tid}; *(R*)(base + __umul24(idx, 12u)) = r; }
            else { uint4 r = {h, (uint32_t)i, (uint32_t)tid, 0u}; *(uint4*)(base + (idx << 4)) = r; }
        }
    }
    __syncthreads();
    if (tid < 256) counts[(size_t)blockIdx.x * 256 + tid] = s_cnt[tid >> 6][tid & 63];
}

// Staged form: every WAVEFRONT owns an LDS ring of RING records per queue (8 wavefronts x Q x RING x 12 B: 96 KB at Q = 64). A
// lane whose ring is full waits; after every record the wavefront writes its full rings out, one ring per iteration, as one
// contiguous burst (RING x 12 bytes; the place in the global queue comes from an LDS counter shared by the two wavefronts of an
// encoding). Only the owning wavefront touches a ring, so no barrier is involved.
__global__ __launch_bounds__(512) void k_staged(char* out, uint32_t* counts, int qshift)
{
    extern __shared__ uint32_t lds[];
    const int nq = 1 << qshift;
    uint32_t* s_done = lds;                               // [4][64] records of the (encoding, queue) already placed in memory
    uint32_t* s_fill = lds + 256;                         // [8][64] records in the wavefront's ring
    uint32_t* s_ring = lds + 256 + 512;                   // [8][nq][RING][3]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), e = wv >> 1;
    if (tid < 256) s_done[tid] = 0;
    s_fill[tid] = 0;
    __syncthreads();
    const int sub_shift = 13 - qshift;
    char* base = out + ((size_t)blockIdx.x * 4 + e) * CT * 12;
    uint32_t* fill = s_fill + wv * 64;
    uint32_t* ring = s_ring + (size_t)wv * nq * RING * 3;
    auto flush = [&](uint32_t q, uint32_t n) {            // whole wavefront: the n records of its ring q, contiguous in memory
        uint32_t done = 0;
        if (lane == 0) { done = atomicAdd(&s_done[e * 64 + q], n); fill[q] = 0; }
        done = (uint32_t)__shfl((int)done, 0, 64);
        if (done + n <= (1u << sub_shift)) {
            uint32_t* dst = (uint32_t*)(base + __umul24((q << sub_shift) + done, 12u));
            for (uint32_t w = lane; w < n * 3; w += 64) dst[w] = ring[q * RING * 3 + w];
        }
    };
    uint32_t h = mix(blockIdx.x * 512u + tid);
    for (int i = 0; i < RECS; ++i) {
        h = mix(h + i);
        const uint32_t q = h & (nq - 1u);
        bool pending = true;
        while (__any(pending)) {
            if (pending) {
                const uint32_t slot = atomicAdd(&fill[q], 1u);
                if (slot < RING) {
                    uint32_t* r = ring + (q * RING + slot) * 3;
                    r[0] = h; r[1] = (uint32_t)i; r[2] = (uint32_t)tid;
                    pending = false;
                }
            }
            unsigned long long full = __ballot(pending);  // lanes that found their ring full: flush those rings, then retry
            while (full) {
                const int l0 = __ffsll((long long)full) - 1;
                const uint32_t qf = (uint32_t)__shfl((int)q, l0, 64);
                flush(qf, RING);
                full &= ~__ballot(pending && q == qf);
            }
        }
    }
    for (int q = 0; q < nq; ++q) {                        // leftovers of this wavefront
        const uint32_t n = min(fill[q], (uint32_t)RING);
        if (n) flush((uint32_t)q, n);
    }
    __syncthreads();
    if (tid < 256) counts[(size_t)blockIdx.x * 256 + tid] = s_done[tid];
}

int main()
{
    const int blocks = 4096;                              // ~ (256 tiles x 16 levels) of a 262 k-sample batch
    char* out; uint32_t* counts;
    (void)hipMalloc(&out, (size_t)blocks * 4 * CT * 16);
    (void)hipMalloc(&counts, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const double recs = (double)blocks * 512 * RECS;
    for (int qshift : {3, 5, 6}) {
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(a);
                if (variant == 0) hipLaunchKernelGGL(k_direct<12>, dim3(blocks), dim3(512), 0, 0, out, counts, qshift);
                else if (variant == 1) hipLaunchKernelGGL(k_direct<16>, dim3(blocks), dim3(512), 0, 0, out, counts, qshift);
                else hipLaunchKernelGGL(k_staged, dim3(blocks), dim3(512), (256 + 512 + 8 * (1 << qshift) * RING * 3) * 4, 0, out, counts, qshift);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            std::vector<uint32_t> c((size_t)blocks * 256);
            (void)hipMemcpy(c.data(), counts, c.size() * 4, hipMemcpyDeviceToHost);
            double written = 0; for (uint32_t v : c) written += v;
            printf("queues %2d  %-9s  %.3f ms  %.2f ns per 1000 records  %.1f G records/s  (counted %.0f of %.0f)\n", 1 << qshift,
                   variant == 0 ? "direct12" : variant == 1 ? "direct16" : "staged12", best, best * 1e9 / recs, recs / best / 1e6, written, recs);
        }
    }
    return 0;
}
