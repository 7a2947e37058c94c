// Non-returning fp32 global atomics on gfx950: rate by memory scope and by how much of the chip issues them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <int SCOPE>
__global__ void k(float* buf, uint32_t mask, int iters)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = rng(s);
        float* p = buf + (s & mask);
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <int SCOPE>
void run(float* buf, uint32_t mask, int blocks, const char* name)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int threads = 256, iters = 256;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<SCOPE>, dim3(blocks), dim3(threads), 0, 0, buf, mask, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    }
    printf("%-40s blocks %5d : %8.3f ms  %7.2f G lane-atomics/s\n", name, blocks, ms, (double)blocks * threads * iters / ms / 1e6);
}
int main()
{
    const size_t big = 64u << 20;
    float* buf; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 0, big * 4);
    const uint32_t m256 = (uint32_t)big - 1, m2 = (512u << 10) - 1;
    for (int blocks : {2048, 256, 64, 8}) {
        run<0>(buf, m2, blocks, "agent scope, 2MB");
        run<1>(buf, m2, blocks, "workgroup scope, 2MB");
        run<2>(buf, m2, blocks, "wavefront scope, 2MB");
        run<3>(buf, m2, blocks, "system scope, 2MB");
        run<0>(buf, m256, blocks, "agent scope, 256MB");
        run<1>(buf, m256, blocks, "workgroup scope, 256MB");
    }
    return 0;
}
