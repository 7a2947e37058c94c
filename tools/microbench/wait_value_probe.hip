// Probe (round 6): can a HIP stream wait for a counter that a RUNNING kernel on another stream advances (hipStreamWaitValue32, >=)?
// What the data-parallel step would use it for: ONE accumulate launch over all temporal segments bumps a per-group counter as its
// workgroups finish a segment; the stream a group's reduce-scatter is issued from waits for "counter >= workgroups of the group" --
// the collective of group g starts while the same launch is still accumulating group g + 1, with no extra launches and no tails.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/wait_value_probe.hip -o tools/microbench/_build/wait_value_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_producer(int32_t* counter, unsigned long long* stamps, int phases, long long spin)
{
    // one workgroup: `phases` phases of ~spin clocks each; after phase p it publishes counter = p + 1
    for (int p = 0; p < phases; ++p) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) { }
        if (threadIdx.x == 0) {
            stamps[p] = wall_clock64();
            __threadfence_system();
            atomicAdd(counter, 1);
        }
        __syncthreads();
    }
}
__global__ void k_consumer(unsigned long long* stamp, const int32_t* counter, int32_t* seen)
{
    if (threadIdx.x == 0) { *stamp = wall_clock64(); *seen = *counter; }
}

int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    int32_t* counter; unsigned long long* stamps; int32_t* seen;
    const int phases = 4;
    CK(hipMalloc(&counter, 64)); CK(hipMalloc(&stamps, 64 * 8)); CK(hipMalloc(&seen, 64));
    CK(hipMemset(counter, 0, 64)); CK(hipMemset(stamps, 0, 64 * 8)); CK(hipMemset(seen, 0, 64));
    hipStream_t a, b[phases];
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    for (int p = 0; p < phases; ++p) CK(hipStreamCreateWithFlags(&b[p], hipStreamNonBlocking));
    // consumers first (they must NOT run until the producer reaches their phase), then the producer: 4 phases of 20 ms at 100 MHz
    for (int p = 0; p < phases; ++p) {
        CK(hipStreamWaitValue32(b[p], counter, (uint32_t)(p + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(k_consumer, dim3(1), dim3(64), 0, b[p], stamps + 8 + p, counter, seen + p);
    }
    hipLaunchKernelGGL(k_producer, dim3(1), dim3(64), 0, a, counter, stamps, phases, 2000000LL);
    CK(hipDeviceSynchronize());
    unsigned long long h[16]; int32_t hs[8];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, seen, sizeof(hs), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int p = 0; p < phases; ++p) {
        const double after = ((double)h[8 + p] - (double)h[p]) / 100.0;          // us after the phase was published (100 MHz clock)
        const double before_next = p + 1 < phases ? ((double)h[p + 1] - (double)h[8 + p]) / 100.0 : 0.0;
        printf("phase %d: consumer ran %.1f us after the producer published it, %.1f us before the next phase ended, saw counter %d\n",
               p, after, before_next, hs[p]);
        if (!(after >= 0.0) || hs[p] < p + 1 || (p + 1 < phases && !(before_next > 0.0))) ok = 0;
    }
    printf(ok ? "OK: each consumer started after its phase and before the producer finished the next one\n" : "NOT OK\n");
    return ok ? 0 : 2;
}
