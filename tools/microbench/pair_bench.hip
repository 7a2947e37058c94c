// Does fetching the x-neighbour corner pair of a hash-grid cell with ONE wide load pay on gfx950? (round 6)
// tcnn's spatial hash is x ^ (y P1) ^ (z P2): the entries of corners (x0, y, z) and (x0 + 1, y, z) are the two halves of one aligned
// 8-byte pair when x0 is even, and unrelated when x0 is odd. Per lane and per "corner pair" the kernel compares
//   A  two 4-byte gathers (what the level body does today),
//   C  one 8-byte gather of the aligned pair of corner x0 by every active lane + one 4-byte gather by the odd-x0 lanes only,
//   D  one 16-byte gather of the aligned quad by every active lane + one 4-byte gather by the lanes with x0 % 4 == 3,
// 16 corner pairs (4 pairs x 4 encodings of one level) in flight per iteration, `active` of the 64 lanes taking part (the head lanes
// of the cell sharing), random entries in a table of `footprint` bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t mask, int stride, int iters, uint32_t* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t s = tid * 2654435761u + 12345u;
    uint32_t acc = 0;
    const bool on = (lane % stride) == 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t i0[16], i1[16];
        s = rng(s);
        const uint32_t x0 = s >> 7;                       // the cell's x coordinate: its parity decides
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            s = rng(s);
            const uint32_t h = s;                         // y, z part of the hash
            i0[u] = (x0 ^ h) & mask;
            i1[u] = ((x0 + 1u) ^ h) & mask;
        }
        if (on) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += buf[i0[u]];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += buf[i1[u]];
            } else if (MODE == 1) {
                uint2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *(const uint2*)(buf + (i0[u] & ~1u));
                uint32_t w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = 0;
                if (x0 & 1u) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) w[u] = buf[i1[u]];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (v[u].x ^ v[u].y) + w[u];
            } else {
                uint4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *(const uint4*)(buf + (i0[u] & ~3u));
                uint32_t w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = 0;
                if ((x0 & 3u) == 3u) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) w[u] = buf[i1[u]];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (v[u].x ^ v[u].y ^ v[u].z ^ v[u].w) + w[u];
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE>
float run(const uint32_t* buf, uint32_t* out, uint32_t mask, int stride)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 16, threads = 256, iters = 64;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, buf, mask, stride, iters, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    }
    return ms;
}
int main()
{
    const size_t big = 64u << 20;
    uint32_t *buf, *out; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 1, big * 4); (void)hipMalloc(&out, 4);
    const uint32_t masks[] = {(4u << 10) - 1, (256u << 10) - 1, (4u << 20) - 1};
    const char* names[] = {"16KB", "1MB", "16MB"};
    printf("# per (wavefront, 16 corner pairs): A two 4-B gathers | C 8-B pair + 4-B for odd x0 | D 16-B quad + 4-B for x0%%4==3\n");
    for (int m = 0; m < 3; ++m)
        for (int stride : {1, 2, 4, 16}) {
            const float a = run<0>(buf, out, masks[m], stride), c = run<1>(buf, out, masks[m], stride), d = run<2>(buf, out, masks[m], stride);
            const double pairs = (double)256 * 16 * 4 * 64 * 16;      // wavefront-level corner pairs
            printf("footprint %-5s active lanes %2d : A %7.3f ms (%5.1f clk/pair/CU)  C %7.3f ms (%5.1f)  D %7.3f ms (%5.1f)\n", names[m],
                   64 / stride, a, 2.4e9 * 256 * a * 1e-3 / pairs, c, 2.4e9 * 256 * c * 1e-3 / pairs, d, 2.4e9 * 256 * d * 1e-3 / pairs);
        }
    return 0;
}
