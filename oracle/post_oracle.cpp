/*
 * oracle/post_oracle.cpp -- TEST INFRASTRUCTURE (CPU oracle, never linked into the product).
 * PARITY UNPINNED (see ptrt_oracle.cpp): the reference ships no vectors for this stage either.
 *
 * Restates the rest of Scene::render_to_device's post chain, literally,
 * launch by launch, including the reference's mip bookkeeping (mip_w/mip_h are halved on the way
 * down and DOUBLED on the way up, so for odd sizes the up-sampling passes address the mips with
 * the doubled sizes, not their true ones -- scene.cuh:1169-1178):
 *   bloom_bright_pass_kernel     src/pathtracer/scene/scene_kernels.cuh:283-299
 *   bloom_blur_h_kernel          scene_kernels.cuh:301-323
 *   bloom_downsample_v_kernel    scene_kernels.cuh:325-352
 *   bloom_upsample_add_kernel    scene_kernels.cuh:354-388
 *   upscale_bilinear_kernel      scene_kernels.cuh:406-441
 *   the chain                    src/pathtracer/scene/scene.cuh:1137-1201, mips allocated :809-822
 *
 * Same arithmetic contract as the path: -ffp-contract=off (a*b+c is two roundings), / IEEE.
 */
#include "detmath.h"

#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float a, float b, float c) { return V3{a, b, c}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator*(V3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
inline V3 operator*(float t, V3 a) { return v3(t * a.x, t * a.y, t * a.z); }
inline V3 lerp(V3 a, V3 b, float t) { return (1.0f - t) * a + t * b; } // common/vec3.cuh:155-157
inline float clampf(float x, float a, float b) { return dm_min(dm_max(x, a), b); }
inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }

const float WEIGHTS[3] = {0.227027f, 0.316216f, 0.070270f};

void bright_pass(V3 *out, const V3 *in, int W, int H, float threshold, float knee) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t idx = (size_t)y * W + x;
            const V3 color = in[idx];
            const float brightness = dm_max(color.x, dm_max(color.y, color.z));
            const float soft_t = brightness - threshold + knee;
            const float bloom = clampf(soft_t / (2.0f * knee) + 0.5f, 0.0f, 1.0f);
            out[idx] = color * bloom;
        }
}

void blur_h(V3 *out, const V3 *in, int W, int H) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t idx = (size_t)y * W + x;
            V3 color = in[idx] * WEIGHTS[0];
            for (int i = 1; i <= 2; ++i) {
                const int x_l = imax(x - i, 0), x_r = imin(x + i, W - 1);
                color = color + in[(size_t)y * W + x_l] * WEIGHTS[i];
                color = color + in[(size_t)y * W + x_r] * WEIGHTS[i];
            }
            out[idx] = color;
        }
}

void downsample_v(V3 *out, const V3 *in, int in_W, int in_H) {
    const int out_W = in_W / 2, out_H = in_H / 2;
    for (int y = 0; y < out_H; ++y)
        for (int x = 0; x < out_W; ++x) {
            const int in_x = x * 2, in_y = y * 2;
            V3 color = v3(0.0f, 0.0f, 0.0f);
            for (int j = -2; j <= 2; ++j) {
                const int tap = imin(imax(in_y + j, 0), in_H - 1);
                color = color + in[(size_t)tap * in_W + in_x] * WEIGHTS[j < 0 ? -j : j];
            }
            out[(size_t)y * out_W + x] = color;
        }
}

void upsample_add(V3 *out_color, const V3 *in_bloom, int W_low, int H_low) {
    const int W_high = W_low * 2, H_high = H_low * 2;
    for (int y = 0; y < H_high; ++y)
        for (int x = 0; x < W_high; ++x) {
            const float u = (x + 0.5f) / (float)W_high;
            const float v = (y + 0.5f) / (float)H_high;
            const float u_low = u * W_low - 0.5f;
            const float v_low = v * H_low - 0.5f;
            int x0 = (int)floorf(u_low), y0 = (int)floorf(v_low);
            const float u_frac = u_low - x0, v_frac = v_low - y0;
            const int x1 = imin(x0 + 1, W_low - 1), y1 = imin(y0 + 1, H_low - 1);
            x0 = imax(x0, 0);
            y0 = imax(y0, 0);
            const V3 s00 = in_bloom[(size_t)y0 * W_low + x0], s10 = in_bloom[(size_t)y0 * W_low + x1];
            const V3 s01 = in_bloom[(size_t)y1 * W_low + x0], s11 = in_bloom[(size_t)y1 * W_low + x1];
            const V3 bloom = lerp(lerp(s00, s10, u_frac), lerp(s01, s11, u_frac), v_frac);
            const size_t idx_high = (size_t)y * W_high + x;
            out_color[idx_high] = out_color[idx_high] + bloom;
        }
}

constexpr int BLOOM_MIP_LEVELS = 6; // scene.cuh:159

} // namespace

extern "C" {

/* Step 5 of render_to_device on `image` (cur_w x cur_h, modified in place).  alloc_w/alloc_h are the
 * FULL frame sizes the mips were allocated from (scene.cuh:809-822).  Returns 0, or -1 where the
 * reference would dereference a NULL mip (a chain level of zero size). */
int oracle_bloom(float *image, int cur_w, int cur_h, int alloc_w, int alloc_h) {
    V3 *img = reinterpret_cast<V3 *>(image);
    std::vector<std::vector<V3>> chain(BLOOM_MIP_LEVELS);
    bool have[BLOOM_MIP_LEVELS];
    int aw = alloc_w, ah = alloc_h;
    for (int i = 0; i < BLOOM_MIP_LEVELS; ++i) {
        aw /= 2;
        ah /= 2;
        have[i] = !(aw == 0 || ah == 0);
        if (have[i])
            chain[i].assign((size_t)aw * ah, v3(0.0f, 0.0f, 0.0f));
    }
    std::vector<V3> bright((size_t)alloc_w * alloc_h), temp((size_t)alloc_w * alloc_h);
    bright_pass(bright.data(), img, cur_w, cur_h, 1.5f, 0.5f);
    int mip_w = cur_w, mip_h = cur_h;
    const V3 *last = bright.data();
    for (int i = 0; i < BLOOM_MIP_LEVELS; ++i) {
        if (!have[i])
            break;
        const int next_w = mip_w / 2, next_h = mip_h / 2;
        if (next_w == 0 || next_h == 0)
            break;
        blur_h(temp.data(), last, mip_w, mip_h);
        downsample_v(chain[i].data(), temp.data(), mip_w, mip_h);
        last = chain[i].data();
        mip_w = next_w;
        mip_h = next_h;
    }
    for (int i = BLOOM_MIP_LEVELS - 2; i >= 0; --i) {
        if (!have[i])
            continue;
        mip_w *= 2;
        mip_h *= 2;
        if (!have[i + 1])
            return -1;
        upsample_add(chain[i].data(), chain[i + 1].data(), mip_w / 2, mip_h / 2);
    }
    if (!have[0])
        return -1;
    upsample_add(img, chain[0].data(), cur_w / 2, cur_h / 2);
    return 0;
}

/* upscale_bilinear_kernel */
void oracle_upscale(float *out_, const float *in_, int out_w, int out_h, int in_w, int in_h) {
    V3 *out = reinterpret_cast<V3 *>(out_);
    const V3 *in = reinterpret_cast<const V3 *>(in_);
    for (int y = 0; y < out_h; ++y)
        for (int x = 0; x < out_w; ++x) {
            float u = (x + 0.5f) * (float)in_w / (float)out_w - 0.5f;
            float v = (y + 0.5f) * (float)in_h / (float)out_h - 0.5f;
            u = dm_max(0.0f, dm_min((float)(in_w - 1), u));
            v = dm_max(0.0f, dm_min((float)(in_h - 1), v));
            const int x0 = (int)floorf(u), y0 = (int)floorf(v);
            const int x1 = imin(x0 + 1, in_w - 1), y1 = imin(y0 + 1, in_h - 1);
            const float fx = u - x0, fy = v - y0;
            const V3 s00 = in[(size_t)y0 * in_w + x0], s10 = in[(size_t)y0 * in_w + x1];
            const V3 s01 = in[(size_t)y1 * in_w + x0], s11 = in[(size_t)y1 * in_w + x1];
            const V3 top = s00 * (1.0f - fx) + s10 * fx;
            const V3 bot = s01 * (1.0f - fx) + s11 * fx;
            out[(size_t)y * out_w + x] = top * (1.0f - fy) + bot * fy;
        }
}

} // extern "C"
