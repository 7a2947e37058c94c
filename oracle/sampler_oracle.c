/*
 * oracle/sampler_oracle.c -- CPU restatement of the reference ray sampler. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 * The product path (humanrf_amd/) never imports, links or executes anything under oracle/.
 *
 * PARITY (round 5): pinned bit for bit against the reference's own kernel source compiled for the host
 * (oracle/build_ref_cuda.py, tests/test_cpu_ref_cuda.py) under the floating-point semantics fixed below; the
 * CUDA texture unit and nvcc's --use_fast_math stay definitions: the reference ships no tests / golden vectors and
 * its .cu files cannot be built as they are
 * (nvcc + CUDA texture hardware + GLM). This file restates, line by line,
 *   actorshq/dataset/native/ray_sampler.cu:11-26    compute_aabb_minmax
 *   actorshq/dataset/native/ray_sampler.cu:28-78    compute_occupancy_minmax
 *   actorshq/dataset/native/ray_sampler.cu:96-146   compute_minmax_kernel (pixel -> ray)
 *   actorshq/dataset/native/ray_sampler.cu:165-193  compute_sample_distances_kernel
 *   actorshq/dataset/native/occupancy_grid.cu:28-35 texture descriptor (trilinear, clamp,
 *                                                   normalized coords, normalized-float read)
 * with the floating-point semantics the build FIXES, because the
 * reference's own (--use_fast_math + texture unit) are not reproducible across vendors:
 *   - IEEE fp32, round-to-nearest, no FMA contraction (compile with -ffp-contract=off, no fast-math)
 *   - inv_dir = 1.0f / dir; normalize(v) = v * (1.0f / sqrtf(x*x + y*y + z*z))
 *   - mat3 * vec3 = m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z (column-major, left to right)
 *   - glm::min(a,b) = (b < a) ? b : a ; glm::max(a,b) = (a < b) ? b : a
 *   - texture fetch: xB = x*G - 0.5; i = floor(xB); a = xB - i; aq = floor(a*256 + 0.5) (9-bit
 *     fixed point, 8 fractional bits); texel indices clamped to [0,G-1]. The sampler only ever
 *     tests "tex > 0", which for u8 texels is: some tap with non-zero quantised weight is non-zero.
 *   - count = (int)((tmax - tmin) / step)  (true division, C truncation)
 *   - t = tmin + (float)local * step       (mul, then add)
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }

/* "tex3D<float>(grid, x, y, z) > 0" for a (G,G,G) uint8 volume laid out [z][y][x]. */
static int tex_gt0(const uint8_t* g, int G, float x, float y, float z)
{
    const float c[3] = {x, y, z};
    int i0[3], i1[3], a0[3], a1[3];
    for (int d = 0; d < 3; ++d) {
        float xb = c[d] * (float)G - 0.5f;
        float fl = floorf(xb);
        float fr = xb - fl;
        int aq = (int)floorf(fr * 256.0f + 0.5f);
        /* clamp in float domain first so that absurd coordinates cannot overflow the int */
        float flc = fl < -1.0f ? -1.0f : (fl > (float)G ? (float)G : fl);
        if (!(fl == fl)) { flc = -1.0f; aq = 0; } /* NaN coordinate: samples texel 0 only */
        int i = (int)flc;
        int lo = i < 0 ? 0 : (i > G - 1 ? G - 1 : i);
        int hi = (i + 1) < 0 ? 0 : ((i + 1) > G - 1 ? G - 1 : (i + 1));
        i0[d] = lo; i1[d] = hi;
        a0[d] = aq < 256;  /* lower tap has weight (256-aq)/256 */
        a1[d] = aq > 0;    /* upper tap has weight aq/256 */
    }
    for (int cz = 0; cz < 2; ++cz) {
        if (!(cz ? a1[2] : a0[2])) continue;
        size_t oz = (size_t)(cz ? i1[2] : i0[2]) * G * G;
        for (int cy = 0; cy < 2; ++cy) {
            if (!(cy ? a1[1] : a0[1])) continue;
            size_t oy = oz + (size_t)(cy ? i1[1] : i0[1]) * G;
            for (int cx = 0; cx < 2; ++cx) {
                if (!(cx ? a1[0] : a0[0])) continue;
                if (g[oy + (size_t)(cx ? i1[0] : i0[0])] > 0) return 1;
            }
        }
    }
    return 0;
}

static inline int occ_at(const uint8_t* g, int G, const float o[3], const float d[3], float t)
{
    /* current_point = ray_origin + ray_direction * t + 0.5f   (ray_sampler.cu:39,54,69,187) */
    float px = (o[0] + d[0] * t) + 0.5f;
    float py = (o[1] + d[1] * t) + 0.5f;
    float pz = (o[2] + d[2] * t) + 0.5f;
    return tex_gt0(g, G, px, py, pz);
}

/* ray_sampler.cu:11-26 */
static void aabb_minmax(const float aabb[6], const float o[3], const float d[3], float mm[2])
{
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
        float inv = 1.0f / d[k];
        float t0 = (aabb[k] - o[k]) * inv;
        float t1 = (aabb[3 + k] - o[k]) * inv;
        mn[k] = gmin(t0, t1);
        mx[k] = gmax(t0, t1);
    }
    mm[0] = gmax(mn[0], gmax(mn[1], mn[2]));
    mm[1] = gmin(mx[0], gmin(mx[1], mx[2]));
}

/* ray_sampler.cu:28-78 */
static void occupancy_minmax(const float aabb[6], const float o[3], const float d[3],
                             const uint8_t* g, int G, float step, float out[2])
{
    float mm[2];
    aabb_minmax(aabb, o, d, mm);
    float tmin = mm[0];
    while (tmin < mm[1]) {
        if (occ_at(g, G, o, d, tmin)) break;
        tmin += step;
    }
    if (tmin < mm[1]) {
        float refine = -step * 0.5f;
        for (int i = 0; i < 5; ++i) {
            tmin += refine;
            if (occ_at(g, G, o, d, tmin)) refine = -fabsf(refine) * 0.5f;
            else refine = fabsf(refine) * 0.5f;
        }
    }
    float tmax = mm[1];
    while (tmax > tmin) {
        if (occ_at(g, G, o, d, tmax)) break;
        tmax -= step;
    }
    out[0] = tmin;
    out[1] = tmax;
}

/*
 * compute_minmax_kernel<kOccupancyMinmax> for every requested ray (ray_sampler.cu:80-147).
 * grids: array of B pointers to (G,G,G) u8 volumes (NULL array when use_occupancy == 0).
 */
void orc_minmax(const float* inverse_krs, const float* camera_origins, const uint8_t* landscape,
                const int64_t* ray_indices, const uint8_t* const* grids, const float* aabb,
                int64_t num_rays, int grid_resolution, int width_in, int height_in, int use_occupancy,
                float* out_dirs, float* out_minmax, uint8_t* out_mask)
{
    for (int64_t r = 0; r < num_rays; ++r) {
        int width = width_in, height = height_in;
        const int64_t idx = ray_indices[r];
        const int image = (int)(idx / ((int64_t)width * height));
        if (!landscape[image]) { int t = width; width = height; height = t; }
        const float px = (float)(idx % width) + 0.5f;
        const float py = (float)((idx / width) % height) + 0.5f;
        const float* m = inverse_krs + (size_t)image * 9; /* column-major: m[col*3+row] */
        const float* o = camera_origins + (size_t)image * 3;
        float v[3];
        for (int k = 0; k < 3; ++k) v[k] = (m[0 + k] * px + m[3 + k] * py) + m[6 + k] * 1.0f;
        float dot = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
        float inv = 1.0f / sqrtf(dot);
        float d[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
        float mm[2];
        if (use_occupancy) occupancy_minmax(aabb, o, d, grids[image], grid_resolution, 0.5f / (float)grid_resolution, mm);
        else aabb_minmax(aabb, o, d, mm);
        out_dirs[r * 3 + 0] = d[0]; out_dirs[r * 3 + 1] = d[1]; out_dirs[r * 3 + 2] = d[2];
        out_minmax[r * 2 + 0] = mm[0]; out_minmax[r * 2 + 1] = mm[1];
        out_mask[r] = mm[0] < mm[1];
    }
}

/* ((tmax - tmin) / step).to(int)  (ray_sampler.cu:283-285) */
void orc_counts(const float* minmax, int64_t num_rays, float step, int32_t* out_counts)
{
    for (int64_t r = 0; r < num_rays; ++r)
        out_counts[r] = (int32_t)((minmax[r * 2 + 1] - minmax[r * 2 + 0]) / step);
}

/*
 * compute_sample_distances_kernel (ray_sampler.cu:149-194) over the COMPACTED rays, followed by the
 * boolean-mask compaction of ray_sampler.cu:322-323. Returns the number of surviving samples.
 * grids[r] is the volume of the image ray r came from (NULL array: aabb mode, keep everything).
 */
int64_t orc_samples(const float* origins, const float* dirs, const float* minmax, const int32_t* counts,
                    const uint8_t* const* grids, int64_t num_rays, int grid_resolution, float step,
                    float* out_t, int32_t* out_ray)
{
    int64_t n = 0;
    for (int64_t r = 0; r < num_rays; ++r) {
        for (int32_t local = 0; local < counts[r]; ++local) {
            float t = minmax[r * 2] + (float)local * step;
            int keep = 1;
            if (grids) keep = occ_at(grids[r], grid_resolution, origins + r * 3, dirs + r * 3, t);
            if (keep) {
                if (out_t) { out_t[n] = t; out_ray[n] = (int32_t)r; }
                ++n;
            }
        }
    }
    return n;
}

/* single texture predicate, exposed for known-answer tests */
int orc_tex_gt0(const uint8_t* g, int G, float x, float y, float z) { return tex_gt0(g, G, x, y, z); }

/*
 * nerfacc 0.3.1 render_visibility as called at humanrf/volume_rendering.py:75-81
 * [UPSTREAM-KNOWLEDGE, SURVEY.md A.4]: per ray, T_i = prod_{j<i} (1 - alpha_j) (exclusive, every
 * sample contributes), visible_i = (T_i >= early_stop_eps) && (alpha_i >= alpha_thre).
 * The build FIXES the product order: sequential fp32 multiplication in sample order.
 * ray_indices must be sorted ascending (contiguous run per ray), as the sampler produces them.
 */
void orc_visibility(const float* alphas, const int64_t* ray_indices, int64_t n,
                    float early_stop_eps, float alpha_thre, uint8_t* out_vis)
{
    int64_t i = 0;
    while (i < n) {
        const int64_t ray = ray_indices[i];
        float T = 1.0f;
        for (; i < n && ray_indices[i] == ray; ++i) {
            const float a = alphas[i];
            out_vis[i] = (T >= early_stop_eps) && (a >= alpha_thre);
            T = T * (1.0f - a);
        }
    }
}
