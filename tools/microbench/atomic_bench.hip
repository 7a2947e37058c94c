// Micro-benchmark: throughput of non-returning fp32 / packed-fp16 global atomics on gfx950 by address pattern.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
// group: number of adjacent lanes that share one contiguous group of dwords; mode 0 fp32, 1 half2
template <int MODE>
__global__ void k(float* buf, uint32_t mask_entries, int group, int iters)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid / group * 2654435761u + 12345u;
    const uint32_t sub = tid % group;
    for (int i = 0; i < iters; ++i) {
        s = rng(s);
        // base aligned to `group` dwords
        const uint32_t base = (s & mask_entries) / group * group;
        if (MODE == 0) unsafeAtomicAdd(buf + base + sub, 1.0f);
        else unsafeAtomicAdd((__half2*)buf + base + sub, __floats2half2_rn(1.0f, 1.0f));
    }
}
int main()
{
    const size_t big = 64u << 20;  // dwords -> 256 MB
    float* buf; hipMalloc(&buf, big * 4); hipMemset(buf, 0, big * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, threads = 256, iters = 256;
    struct Cfg { const char* name; uint32_t mask; int group; int mode; };
    std::vector<Cfg> cfgs = {
        {"fp32 random, 256MB, group1", (uint32_t)big - 1, 1, 0}, {"fp32 random, 256MB, group2", (uint32_t)big - 1, 2, 0},
        {"fp32 random, 256MB, group4", (uint32_t)big - 1, 4, 0}, {"fp32 random, 256MB, group16", (uint32_t)big - 1, 16, 0},
        {"fp32 random, 256MB, group32", (uint32_t)big - 1, 32, 0},
        {"fp32 random, 32MB, group1", (8u << 20) - 1, 1, 0}, {"fp32 random, 32MB, group2", (8u << 20) - 1, 2, 0},
        {"fp32 random, 2MB, group1", (512u << 10) - 1, 1, 0}, {"fp32 random, 2MB, group2", (512u << 10) - 1, 2, 0},
        {"fp32 random, 2MB, group4", (512u << 10) - 1, 4, 0}, {"fp32 random, 2MB, group16", (512u << 10) - 1, 16, 0},
        {"half2 random, 256MB, group1", (uint32_t)big - 1, 1, 1}, {"half2 random, 2MB, group1", (512u << 10) - 1, 1, 1},
        {"half2 random, 256MB, group4", (uint32_t)big - 1, 4, 1},
        {"fp32 same 64 addrs", 63, 1, 0},
    };
    for (auto& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (c.mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, buf, c.mask, c.group, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, buf, c.mask, c.group, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 1) printf("%-32s %8.3f ms  %7.1f G lane-atomics/s\n", c.name, ms, (double)blocks * threads * iters / ms / 1e6);
        }
    }
    return 0;
}
