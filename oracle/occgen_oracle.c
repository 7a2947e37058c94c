/*
 * oracle/occgen_oracle.c -- CPU restatement of the reference's occupancy-grid generation. TEST INFRASTRUCTURE ONLY
 * (tests/ only; the product path never imports, links or executes anything under oracle/).
 *
 * PARITY (round 5): orc_grid_from_masks is pinned bit for bit against the reference's own generate_from_masks_kernel, compiled
 * for the host from the source under /root/reference (oracle/build_ref_cuda.py, tests/test_cpu_ref_cuda.py), under the arithmetic
 * fixed below; the cv2.dilate restatement is pinned against scipy's grey dilation (tests/test_oracle_kat.py). The reference ships no
 * tests / golden vectors and its files cannot be built as they are (nvcc + GLM + OpenCV). Restates, line by line,
 *   actorshq/toolbox/native/occupancy_grid_generation.cu:16-80   generate_from_masks_kernel
 *   actorshq/toolbox/generate_occupancy_grids_from_masks.py:64-77 cv2.dilate(mask, ones((k,k)), iterations=1)
 * with the arithmetic the build FIXES (the reference compiles with --use_fast_math): IEEE fp32, no FMA contraction
 * (-ffp-contract=off), GLM's mat4 * vec4 operand order ((m0*x + m1*y) + (m2*z + m3*w)), true division, float -> int
 * truncating toward zero, saturating, NaN -> 0 (the behaviour of cvt.rzi.s32.f32 and of v_cvt_i32_f32).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static int f2i_rz(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

void orc_grid_from_masks(const uint8_t* masks, const float* proj /* (C,16) column-major */, const uint8_t* landscape,
                         int threshold, int num_cameras, int G, int width, int height, uint8_t* grid /* [z][y][x] */)
{
    const size_t P = (size_t)width * (size_t)height;
    const float den = (float)(G - 1);
    for (int gz = 0; gz < G; ++gz)
        for (int gy = 0; gy < G; ++gy)
            for (int gx = 0; gx < G; ++gx) {
                const float vx = (float)gx / den - 0.5f, vy = (float)gy / den - 0.5f, vz = (float)gz / den - 0.5f;
                int covered = 0, in_hull = 0;
                for (int c = 0; c < num_cameras; ++c) {
                    const float* m = proj + (size_t)c * 16;
                    const int land = landscape[c] != 0;
                    const int cw = land ? width : height, ch = land ? height : width;
                    const float px = (m[0] * vx + m[4] * vy) + (m[8] * vz + m[12] * 1.0f);
                    const float py = (m[1] * vx + m[5] * vy) + (m[9] * vz + m[13] * 1.0f);
                    const float pz = (m[2] * vx + m[6] * vy) + (m[10] * vz + m[14] * 1.0f);
                    const int x = f2i_rz(px / pz), y = f2i_rz(py / pz);
                    if (x >= 0 && x < cw && y >= 0 && y < ch) {
                        const int x1 = (x + 1 < cw - 1) ? x + 1 : cw - 1, y1 = (y + 1 < ch - 1) ? y + 1 : ch - 1;
                        const uint8_t* mk = masks + (size_t)c * P;
                        if (mk[(size_t)x + (size_t)y * cw] == 0 && mk[(size_t)x1 + (size_t)y * cw] == 0 &&
                            mk[(size_t)x + (size_t)y1 * cw] == 0 && mk[(size_t)x1 + (size_t)y1 * cw] == 0) {
                            const int rest = num_cameras - c - 1;
                            if (covered + rest < threshold) break;
                        } else {
                            ++covered;
                            in_hull = covered >= threshold;
                            if (in_hull) break;
                        }
                    }
                }
                grid[((size_t)gz * G + gy) * G + gx] = in_hull ? 255 : 0;
            }
}

void orc_mask_dilate(const uint8_t* in, int width, int height, int k, int64_t images, uint8_t* out)
{
    const int a = k / 2;
    const size_t P = (size_t)width * (size_t)height;
    for (int64_t img = 0; img < images; ++img)
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                uint8_t v = 0;
                for (int dy = -a; dy < k - a; ++dy) {
                    const int yy = y + dy;
                    if (yy < 0 || yy >= height) continue;
                    for (int dx = -a; dx < k - a; ++dx) {
                        const int xx = x + dx;
                        if (xx < 0 || xx >= width) continue;
                        const uint8_t s = in[(size_t)img * P + (size_t)yy * width + xx];
                        if (s > v) v = s;
                    }
                }
                out[(size_t)img * P + (size_t)y * width + x] = v;
            }
}
