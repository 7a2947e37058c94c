// Does a 16x16x16 f16 MFMA accumulate correctly onto the result of a 16x16x32 one (gfx950)? csrc/mlp_common.h pairs an odd
// 16-deep block with a zero fragment in a second 16x16x32 instead, because the mixed chain "gave wrong sums" in round 3
// (every camera_embedding_dim = 2 parity test failed). This is the chain in isolation: D = A[16x48] . B[48x16], blocks 0 and 1
// through v_mfma_f32_16x16x32_f16 (the two 4-element fragments of lane (g, c) concatenated, as Prec<false>::mfma2 does), block 2
// through v_mfma_f32_16x16x16_f16 on the same accumulator, against the same sums taken on the host in double precision.
// build + run: hipcc --offload-arch=gfx950 -O3 mfma_chain_repro.hip -o /tmp/mfma_chain_repro && /tmp/mfma_chain_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// A row-major (16, 48), B row-major (48, 16); lane (g = lane / 16, c = lane % 16) holds A[c][16 b + 4 g + r] and B[16 b + 4 g + r][c]
__global__ void k(const _Float16* A, const _Float16* B, float* D_mixed, float* D_padded)
{
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    h4 a[3], b[3];
    for (int blk = 0; blk < 3; ++blk)
        for (int r = 0; r < 4; ++r) { a[blk][r] = A[c * 48 + 16 * blk + 4 * g + r]; b[blk][r] = B[(16 * blk + 4 * g + r) * 16 + c]; }
    const f4 zero = {0, 0, 0, 0};
    const h4 hz = {0, 0, 0, 0};
    f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7),
                                                    __builtin_shufflevector(b[0], b[1], 0, 1, 2, 3, 4, 5, 6, 7), zero, 0, 0, 0);
    f4 mixed = __builtin_amdgcn_mfma_f32_16x16x16f16(a[2], b[2], acc, 0, 0, 0);                      // 16x16x16 onto a 16x16x32 result
    f4 padded = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a[2], a[2], 0, 1, 2, 3, 4, 5, 6, 7),
                                                       __builtin_shufflevector(b[2], hz, 0, 1, 2, 3, 4, 5, 6, 7), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { D_mixed[(4 * g + r) * 16 + c] = mixed[r]; D_padded[(4 * g + r) * 16 + c] = padded[r]; }
}

int main()
{
    _Float16 hA[16 * 48], hB[48 * 16];
    srand(7);
    for (int i = 0; i < 16 * 48; ++i) { hA[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f); hB[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f); }
    _Float16 *dA, *dB; float *dM, *dP;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dM, 1024); (void)hipMalloc(&dP, 1024);
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dM, dP);
    float M[256], P[256];
    (void)hipMemcpy(M, dM, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(P, dP, 1024, hipMemcpyDeviceToHost);
    double em = 0, ep = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int kk = 0; kk < 48; ++kk) s += (double)hA[i * 48 + kk] * (double)hB[kk * 16 + j];
            em = fmax(em, fabs(M[i * 16 + j] - s)); ep = fmax(ep, fabs(P[i * 16 + j] - s));
        }
    printf("max |D - A.B| over the 16x16 tile: 16x16x32 then 16x16x16 (mixed chain) %.3g   16x16x32 then zero-padded 16x16x32 %.3g\n", em, ep);
    printf("%s\n", em < 1e-3 ? "the mixed chain is CORRECT here: the round-3 failure was not the instruction pair" : "the mixed chain is WRONG");
    return 0;
}
