// Micro-benchmark: throughput of random 4-byte gathers on gfx950 by footprint and by lanes-per-line grouping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
// each lane issues UNROLL independent loads per iteration; `group` adjacent lanes read adjacent dwords
template <int UNROLL>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t mask, int group, int iters, uint32_t* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = (tid / group) * 2654435761u + 12345u;
    const uint32_t sub = tid % group;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t idx[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { s = rng(s); idx[u] = ((s & mask) / group) * group + sub; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += buf[idx[u]];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t big = 64u << 20;  // dwords = 256 MB
    uint32_t *buf, *out; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 1, big * 4); (void)hipMalloc(&out, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int threads = 256, iters = 64;
    struct Cfg { const char* name; uint32_t mask; int group; int blocks; };
    std::vector<Cfg> cfgs;
    const uint32_t sizes[] = {(4u << 10) - 1, (256u << 10) - 1, (8u << 20) - 1, (uint32_t)big - 1};
    const char* names[] = {"16KB", "1MB", "32MB", "256MB"};
    for (int si = 0; si < 4; ++si)
        for (int g : {1, 2, 4, 16})
            cfgs.push_back({names[si], sizes[si], g, 256 * 16});
    for (auto& c : cfgs) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k<32>, dim3(c.blocks), dim3(threads), 0, 0, buf, c.mask, c.group, iters, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double lanes = (double)c.blocks * threads * iters * 32;
        printf("footprint %-6s lanes/line %2d : %8.3f ms  %8.1f G lane-loads/s  %7.1f G lines/s  (%.2f lines/clk/CU @2.4GHz)\n", c.name,
               c.group, ms, lanes / ms / 1e6, lanes / c.group / ms / 1e6, lanes / c.group / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}
