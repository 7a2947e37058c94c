// Does a gather instruction with only K active lanes cost K/64 of a full one? (L1-resident and 256 MB footprints)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t mask, int active, int iters, uint32_t* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t s = tid * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t idx[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { s = rng(s); idx[u] = s & mask; }
        if (lane < active) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += buf[idx[u]];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t big = 64u << 20;
    uint32_t *buf, *out; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 1, big * 4); (void)hipMalloc(&out, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 16, threads = 256, iters = 64;
    const uint32_t masks[] = {(4u << 10) - 1, (256u << 10) - 1, (uint32_t)big - 1};
    const char* names[] = {"16KB", "1MB", "256MB"};
    for (int m = 0; m < 3; ++m)
        for (int act : {64, 32, 16, 8, 4}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, masks[m], act, iters, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
            }
            const double instr = (double)blocks * (threads / 64) * iters * 32;
            printf("footprint %-6s active lanes %2d : %8.3f ms  %7.2f G wave-instr/s  %8.1f G lane-loads/s\n", names[m], act, ms,
                   instr / ms / 1e6, instr * act / ms / 1e6);
        }
    return 0;
}
