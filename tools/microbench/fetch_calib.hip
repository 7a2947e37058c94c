// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the scatter kernels:
// known byte counts, footprint far beyond the 256 MB Infinity Cache, one kernel per pattern.
//   k_read_x1 / x3 / x4   coalesced streaming reads, 4 / 12 / 16 bytes per lane (x3 = k_scatter_accumulate's 768-byte record windows)
//   k_read_gather4        random 4-byte gathers, one 64-byte line each (the calibration of round 1, repeated)
//   k_write_x3 / x4       coalesced streaming writes, 12 / 16 bytes per lane
//   k_write_scatter12     12-byte records to random 12-byte slots (k_scatter_emit's record stores)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); the program prints the bytes
// every kernel moved, the pmc summary (tools/pmc_summary.py) divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct R3 { uint32_t a, b, c; };
__global__ void k_read_x1(const uint32_t* p, size_t n, uint32_t* out) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 0x12345678u) out[0] = s; }
__global__ void k_read_x3(const R3* p, size_t n, uint32_t* out) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { R3 r = p[i]; s += r.a ^ r.b ^ r.c; } if (s == 0x12345678u) out[0] = s; }
__global__ void k_read_x4(const uint4* p, size_t n, uint32_t* out) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 r = p[i]; s += r.x ^ r.y ^ r.z ^ r.w; } if (s == 0x12345678u) out[0] = s; }
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
__global__ void k_read_gather4(const uint32_t* p, uint32_t mask, int per_thread, uint32_t* out) { uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0; for (int i = 0; i < per_thread; ++i) { s = rng(s); acc += p[(s & mask) & ~15u]; } if (acc == 0x12345678u) out[0] = acc; }
__global__ void k_write_x3(R3* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = R3{(uint32_t)i, 1u, 2u}; }
__global__ void k_write_x4(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u); }
__global__ void k_write_scatter12(R3* p, uint32_t mask, int per_thread) { uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 999u; for (int i = 0; i < per_thread; ++i) { s = rng(s); p[s & mask] = R3{s, 1u, 2u}; } }
int main()
{
    const size_t bytes = (size_t)3 << 30;       // 3 GB
    void *buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    const int blocks = 256 * 8, threads = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read_x1, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)buf, bytes / 4, out);
        hipLaunchKernelGGL(k_read_x3, dim3(blocks), dim3(threads), 0, 0, (const R3*)buf, bytes / 12, out);
        hipLaunchKernelGGL(k_read_x4, dim3(blocks), dim3(threads), 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_read_gather4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)buf, (uint32_t)((2u << 30) / 4 - 1), 64, out);
        hipLaunchKernelGGL(k_write_x3, dim3(blocks), dim3(threads), 0, 0, (R3*)buf, bytes / 12);
        hipLaunchKernelGGL(k_write_x4, dim3(blocks), dim3(threads), 0, 0, (uint4*)buf, bytes / 16);
        hipLaunchKernelGGL(k_write_scatter12, dim3(blocks), dim3(threads), 0, 0, (R3*)buf, (uint32_t)((1u << 27) - 1), 64);
    }
    (void)hipDeviceSynchronize();
    const double lines = (double)blocks * threads * 64;
    printf("CALIB k_read_x1 %.0f\nCALIB k_read_x3 %.0f\nCALIB k_read_x4 %.0f\nCALIB k_read_gather4 %.0f (lines x 64 B)\n", (double)bytes, (double)(bytes / 12 * 12), (double)bytes, lines * 64);
    printf("CALIB k_write_x3 %.0f\nCALIB k_write_x4 %.0f\nCALIB k_write_scatter12 %.0f (records x 12 B)\n", (double)(bytes / 12 * 12), (double)bytes, lines * 12);
    return 0;
}
