// TA cost model on gfx950: per-lane load width (4/8/16 B) x active-lane pattern, L1- and L2-resident random gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <int W, int PAT>  // W = dwords per lane; PAT 0 all lanes, 1 lane%4==0, 2 lane<16
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t mask, int iters, uint32_t* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t s = tid * 2654435761u + 12345u;
    uint32_t acc = 0;
    const bool on = PAT == 0 ? true : (PAT == 1 ? (lane & 3) == 0 : lane < 16);
    for (int i = 0; i < iters; ++i) {
        uint32_t idx[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { s = rng(s); idx[u] = (s & mask) & ~(uint32_t)(W - 1); }
        if (on) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (W == 1) acc += buf[idx[u]];
                else if (W == 2) { uint2 v = *(const uint2*)(buf + idx[u]); acc += v.x ^ v.y; }
                else { uint4 v = *(const uint4*)(buf + idx[u]); acc += v.x ^ v.y ^ v.z ^ v.w; }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W, int PAT>
void run(const uint32_t* buf, uint32_t* out, uint32_t mask, const char* fp)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 16, threads = 256, iters = 64;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<W, PAT>), dim3(blocks), dim3(threads), 0, 0, buf, mask, iters, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    }
    const double instr = (double)blocks * (threads / 64) * iters * 16;
    const char* pn[] = {"all 64 lanes", "every 4th lane", "lanes 0..15"};
    printf("footprint %-5s width %2d B  %-14s : %7.3f ms  %6.2f G wave-instr/s  (%5.1f clk/instr/CU @2.4GHz)\n", fp, W * 4, pn[PAT], ms,
           instr / ms / 1e6, 2.4e9 * 256 / (instr / ms * 1e3));
}
int main()
{
    const size_t big = 64u << 20;
    uint32_t *buf, *out; (void)hipMalloc(&buf, big * 4); (void)hipMemset(buf, 1, big * 4); (void)hipMalloc(&out, 4);
    const uint32_t m16k = (4u << 10) - 1, m1m = (256u << 10) - 1;
#define ROW(W) run<W,0>(buf,out,m16k,"16KB"); run<W,1>(buf,out,m16k,"16KB"); run<W,2>(buf,out,m16k,"16KB"); \
               run<W,0>(buf,out,m1m,"1MB"); run<W,1>(buf,out,m1m,"1MB"); run<W,2>(buf,out,m1m,"1MB");
    ROW(1) ROW(2) ROW(4)
    return 0;
}
