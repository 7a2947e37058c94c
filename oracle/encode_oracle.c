/* oracle/encode_oracle.c -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/README.md): an OpenMP restatement of the oracle's own
 * Decomposition4D forward (oracle/hrf_oracle.py: hashgrid_indices, hashgrid_encode(accumulate="fp32"), vectors_sample,
 * compose_tensors), so that the CPU baseline of bench.py can use every host core for the hash gather of the pruning pass
 * ("optional OpenMP C++ for the gather"; torch's intra-op pool degrades beyond ~16 threads on these index ops).
 * It follows, statement by statement, the reference code the Python oracle follows:
 *   humanrf/scene_representation/decomposition4d.py:124-135   Decomposition4D.forward: four encodings + compose
 *   humanrf/scene_representation/native/tensor_composition.cu:37-54   vector taps (coord * Rv - 0.5, clamped) and the composition
 *   tiny-cuda-nn HashGrid [UPSTREAM-KNOWLEDGE, SURVEY.md A.1]: pos = fmaf(x, scale, 0.5), corner order, hash primes, index % size
 * PARITY: pinned bit for bit against oracle.hrf_oracle.decomposition4d (tests/test_oracle_kat.py), which is itself pinned against the
 * reference's Python executed here (tests/test_cpu_ref_fixtures.py). Never linked, imported or called by the product path.
 * Arithmetic: IEEE fp32, no FMA contraction (-ffp-contract=off); the one fused step (pos) is evaluated in double, where the product
 * of two floats is exact, and rounded once -- as the Python oracle does; outputs rounded to half (nearest even) and returned as fp32. */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    float scale;
    int32_t res;
    int64_t size;
    int64_t offset;
    int32_t hashed;
} orc_level;

static float orc_round_half(float x)
{
    /* round to nearest even fp16, back to fp32 (values beyond the half range become inf, like torch's .half()) */
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    float r;
    if (a >= 0x7F800000u) return x;                       /* inf / nan */
    if (a >= 0x477FF000u) {                                /* >= 65520: rounds to inf */
        const uint32_t inf = sign | 0x7F800000u;
        memcpy(&r, &inf, 4);
        return r;
    }
    if (a < 0x38800000u) {                                 /* below 2^-14: half subnormals, spacing 2^-24 */
        const float ax = fabsf(x);
        const float q = ax * 16777216.0f;                  /* exact scaling by 2^24 */
        const float n = nearbyintf(q);                     /* ties to even (default rounding mode) */
        r = n * (1.0f / 16777216.0f);
        return sign ? -r : r;
    }
    {   /* normal: keep 10 mantissa bits, ties to even */
        const uint32_t lsb = (a >> 13) & 1u;
        a += 0x0FFFu + lsb;
        a &= 0xFFFFE000u;
        a |= sign;
        memcpy(&r, &a, 4);
        return r;
    }
}

static void orc_vec_tap(float c, int Rv, int* c0, int* c1, float* fr)
{
    const float coord = c * (float)Rv - 0.5f;
    const float fl = floorf(coord);
    float lo = fl, hi = fl + 1.0f;
    *fr = coord - fl;
    if (lo < 0.0f) lo = 0.0f;
    if (lo > (float)(Rv - 1)) lo = (float)(Rv - 1);
    if (hi < 0.0f) hi = 0.0f;
    if (hi > (float)(Rv - 1)) hi = (float)(Rv - 1);
    *c0 = (int)lo;
    *c1 = (int)hi;
}

/* xyzt (n,4); tables: four (entries,2) fp32 arrays holding half-representable values; vectors (4,Rv,2L); levels[L];
 * out (n, 2L) fp32 holding half-representable values. threads <= 0: OpenMP's default. */
void orc_decomposition4d_fwd(const float* xyzt, const float* const* tables, const float* vectors, const orc_level* levels,
                             int64_t n, int L, int Rv, float* out, int threads)
{
    static const int AX[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};   /* decomposition4d.py:126-129 */
    static const int PAIR[4] = {3, 2, 0, 1};                                     /* tensor_composition.cu:47-54: xyz*v_t + xyt*v_z + yzt*v_x + xzt*v_y */
    const int F = 2 * L;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) if (threads != 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        const float* q = xyzt + 4 * i;
        int c0[4], c1[4];
        float fr[4];
        for (int v = 0; v < 4; ++v) orc_vec_tap(q[v], Rv, &c0[v], &c1[v], &fr[v]);
        for (int l = 0; l < L; ++l) {
            const orc_level lv = levels[l];
            float enc[4][2];
            for (int e = 0; e < 4; ++e) {
                int64_t g[3];
                float w[3];
                for (int d = 0; d < 3; ++d) {
                    const float pos = (float)((double)q[AX[e][d]] * (double)lv.scale + 0.5);
                    const float fl = floorf(pos);
                    w[d] = pos - fl;
                    g[d] = (int64_t)fl;
                }
                float a0 = 0.0f, a1 = 0.0f;
                const float* tb = tables[e] + 2 * lv.offset;
                for (int c = 0; c < 8; ++c) {
                    float weight = 1.0f;
                    int64_t cc[3];
                    for (int d = 0; d < 3; ++d) {
                        if ((c >> d) & 1) { weight = weight * w[d]; cc[d] = g[d] + 1; }
                        else { weight = weight * (1.0f - w[d]); cc[d] = g[d]; }
                    }
                    int64_t index;
                    if (lv.hashed) {
                        const int64_t h = ((cc[0] * 1) & 0xFFFFFFFFLL) ^ ((cc[1] * 2654435761LL) & 0xFFFFFFFFLL) ^
                                          ((cc[2] * 805459861LL) & 0xFFFFFFFFLL);
                        index = h % lv.size;
                    } else {
                        index = ((cc[0] + cc[1] * lv.res + cc[2] * (int64_t)lv.res * lv.res) & 0xFFFFFFFFLL) % lv.size;
                    }
                    a0 = a0 + weight * tb[2 * index];
                    a1 = a1 + weight * tb[2 * index + 1];
                }
                enc[e][0] = orc_round_half(a0);
                enc[e][1] = orc_round_half(a1);
            }
            for (int f = 0; f < 2; ++f) {
                float sv[4];
                for (int v = 0; v < 4; ++v) {
                    const float v0 = vectors[((int64_t)v * Rv + c0[v]) * F + 2 * l + f];
                    const float v1 = vectors[((int64_t)v * Rv + c1[v]) * F + 2 * l + f];
                    sv[v] = v0 + fr[v] * (v1 - v0);
                }
                const float res = ((enc[0][f] * sv[PAIR[0]] + enc[1][f] * sv[PAIR[1]]) + enc[2][f] * sv[PAIR[2]]) + enc[3][f] * sv[PAIR[3]];
                out[i * F + 2 * l + f] = orc_round_half(res);
            }
        }
    }
}
