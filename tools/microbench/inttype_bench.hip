// Micro-benchmark (gfx950): does the TYPE of an atomic change its rate? Non-returning atomics on random addresses,
// (a) global memory: fp32 add / u32 add / u64 add / fp64 add, 2 MB and 256 MB footprints;
// (b) LDS (64 KB per workgroup, 2 workgroups per CU): fp32 add / u32 add / u64 add / returning u32 add, random and
//     conflict-free addresses. Output: lane-atomics per second chip-wide, and cycles per wave-instruction per CU for LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int T>
__global__ void kg(void* buf, uint32_t mask, int iters)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = rng(s);
        const uint32_t j = s & mask;
        if (T == 0) unsafeAtomicAdd((float*)buf + j, 1.0f);
        else if (T == 1) atomicAdd((unsigned int*)buf + j, 1u);
        else if (T == 2) atomicAdd((unsigned long long*)buf + (j >> 1), 1ull);
        else unsafeAtomicAdd((double*)buf + (j >> 1), 1.0);
    }
}

template <int T, bool RANDOM>
__global__ __launch_bounds__(512) void kl(float* out, int iters)
{
    __shared__ unsigned long long s_mem[8192];   // 64 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s_mem[i] = 0ull;
    __syncthreads();
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 777u;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        s = rng(s);
        const uint32_t j = RANDOM ? (s & 16383u) : ((threadIdx.x + i * 64) & 16383u);   // dword index
        if (T == 0) atomicAdd((float*)s_mem + j, 1.0f);
        else if (T == 1) atomicAdd((unsigned int*)s_mem + j, 1u);
        else if (T == 2) atomicAdd(s_mem + (j >> 1), 1ull);
        else if (T == 3) acc += atomicAdd((unsigned int*)s_mem + j, 1u);
        else if (T == 4) { const float v = ((float*)s_mem)[j]; ((float*)s_mem)[j] = v + 1.0f; }   // plain read-modify-write (racy): the LDS data path alone
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s_mem[1] + (float)acc;
}

int main()
{
    const size_t big = 64u << 20;
    void* buf; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 0, big * 4);
    float* out; (void)hipMalloc(&out, 4096 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    const int blocks = 2048, threads = 256, iters = 256;
#define RUNG(T, MASK, NAME)                                                                                       \
    for (int rep = 0; rep < 2; ++rep) {                                                                          \
        (void)hipEventRecord(a); hipLaunchKernelGGL(kg<T>, dim3(blocks), dim3(threads), 0, 0, buf, MASK, iters);  \
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); }           \
    printf("global %-28s %8.3f ms  %7.1f G lane-atomics/s\n", NAME, ms, (double)blocks * threads * iters / ms / 1e6);
    RUNG(0, (512u << 10) - 1, "fp32 add, 2 MB") RUNG(1, (512u << 10) - 1, "u32 add, 2 MB") RUNG(2, (512u << 10) - 1, "u64 add, 2 MB")
    RUNG(3, (512u << 10) - 1, "fp64 add, 2 MB")
    RUNG(0, (uint32_t)big - 1, "fp32 add, 256 MB") RUNG(1, (uint32_t)big - 1, "u32 add, 256 MB") RUNG(2, (uint32_t)big - 1, "u64 add, 256 MB")
    const int lb = 512, lt = 512, li = 2048;   // 2 workgroups per CU resident (64 KB LDS each)
#define RUNL(T, R, NAME)                                                                                          \
    for (int rep = 0; rep < 2; ++rep) {                                                                          \
        (void)hipEventRecord(a); hipLaunchKernelGGL((kl<T, R>), dim3(lb), dim3(lt), 0, 0, out, li);               \
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); }           \
    printf("LDS    %-28s %8.3f ms  %7.1f G lane-ops/s  %6.1f cycles per wave-instruction per CU (2.4 GHz, 256 CUs)\n", NAME, ms, \
           (double)lb * lt * li / ms / 1e6, ms * 1e-3 * 2.4e9 * 256.0 / ((double)lb * lt / 64 * li));
    RUNL(0, true, "fp32 add, random") RUNL(1, true, "u32 add, random") RUNL(2, true, "u64 add, random") RUNL(3, true, "u32 add returning, random")
    RUNL(4, true, "plain load+store, random")
    RUNL(0, false, "fp32 add, conflict-free") RUNL(1, false, "u32 add, conflict-free") RUNL(2, false, "u64 add, conflict-free")
    return 0;
}
