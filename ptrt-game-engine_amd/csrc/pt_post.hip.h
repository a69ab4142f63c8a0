// pt_post.hip.h -- bloom chain, bilinear upscale, final tonemap (gfx950).  SURVEY 8(f) rank 4.
//
// Reference (file:line of Mark-Rindler/PTRT-game-engine), steps 5-7 of Scene::render_to_device
// (scene/scene.cuh:1137-1208):
//   bloom_bright_pass_kernel     scene/scene_kernels.cuh:283-299
//   bloom_blur_h_kernel          scene_kernels.cuh:301-323
//   bloom_downsample_v_kernel    scene_kernels.cuh:325-352
//   bloom_upsample_add_kernel    scene_kernels.cuh:354-388
//   upscale_bilinear_kernel      scene_kernels.cuh:406-441
//   tonemap_kernel               scene/scene.cuh:2004-2047
//
// The reference issues 1 + 6x2 + 5 + 1 (+1) + 1 = 20-21 launches of which most process a few
// thousand pixels; on MI355X such a launch costs ~4.6 us whatever it does.  Here:
//   * bright pass + horizontal blur + vertical down-sample are ONE kernel per level
//     (bloom_down_kernel): the blur is evaluated only on the even columns the down-sample reads,
//     25 taps per output pixel straight from the previous level (the full-size `bright` and
//     `temp` images -- 2 x 12 B/px written and re-read by the reference -- never exist);
//   * consecutive passes of <= 16 K output pixels (the deep levels down AND back up) run in one
//     single-workgroup kernel with barriers between passes (bloom_small_passes_kernel);
//   * the last up-sample-add is fused with the tonemap (bloom_final_kernel), and the up-scale
//     with the tonemap (upscale_tonemap_kernel).
// 1080p: 9 launches instead of 20.  Per-pixel arithmetic is the reference's, term by term (same
// contract as the path), including its mip bookkeeping: sizes are halved on the way down and
// DOUBLED on the way up, so with odd sizes the up passes address a mip with 2*floor(w/2) columns
// (scene.cuh:1169-1178); BloomPass carries those literal sizes.  Bit-identical to
// oracle/post_oracle.cpp.
#pragma once
#include "pt_device.hip.h"

namespace pt {

PT_DEV f3 ldv(const float *p, size_t i) { return mk3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); }
PT_DEV void stv(float *p, size_t i, f3 v) {
    p[i * 3] = v.x;
    p[i * 3 + 1] = v.y;
    p[i * 3 + 2] = v.z;
}

PT_DEV f3 bright_of(f3 color, float threshold, float knee) {
    const float brightness = max_(color.x, max_(color.y, color.z));
    const float soft_t = brightness - threshold + knee;
    const float bloom = clampf(soft_t / (2.0f * knee) + 0.5f, 0.0f, 1.0f);
    return color * bloom;
}

// out(x,y) of level i from `in` (in_W x in_H): blur_h at column 2x of rows 2y-2..2y+2, then the vertical weights
template <bool BRIGHT> PT_DEV f3 bloom_down_pixel(const float *in, int in_W, int in_H, int x, int y, float threshold, float knee) {
    constexpr float WT[3] = {0.227027f, 0.316216f, 0.070270f};
    const int in_x = x * 2, in_y = y * 2;
    f3 color = mk3(0.0f);
    for (int j = -2; j <= 2; ++j) {
        int tap = in_y + j;
        tap = tap < 0 ? 0 : (tap > in_H - 1 ? in_H - 1 : tap);
        const size_t row = (size_t)tap * in_W;
        f3 c0 = ldv(in, row + in_x);
        if (BRIGHT)
            c0 = bright_of(c0, threshold, knee);
        f3 t = c0 * WT[0];
        for (int i = 1; i <= 2; ++i) {
            const int x_l = in_x - i < 0 ? 0 : in_x - i, x_r = in_x + i > in_W - 1 ? in_W - 1 : in_x + i;
            f3 cl = ldv(in, row + x_l), cr = ldv(in, row + x_r);
            if (BRIGHT) {
                cl = bright_of(cl, threshold, knee);
                cr = bright_of(cr, threshold, knee);
            }
            t = t + cl * WT[i];
            t = t + cr * WT[i];
        }
        color = color + t * WT[j < 0 ? -j : j];
    }
    return color;
}

// bloom sample for high-res pixel (x,y) of a W_low x H_low image (bloom_upsample_add_kernel's bilinear fetch)
PT_DEV f3 bloom_up_sample(const float *in_bloom, int W_low, int H_low, int x, int y) {
    const int W_high = W_low * 2, H_high = H_low * 2;
    const float u = ((float)x + 0.5f) / (float)W_high;
    const float v = ((float)y + 0.5f) / (float)H_high;
    const float u_low = u * (float)W_low - 0.5f;
    const float v_low = v * (float)H_low - 0.5f;
    int x0 = (int)__builtin_floorf(u_low), y0 = (int)__builtin_floorf(v_low);
    const float u_frac = u_low - (float)x0, v_frac = v_low - (float)y0;
    const int x1 = x0 + 1 < W_low - 1 ? x0 + 1 : W_low - 1, y1 = y0 + 1 < H_low - 1 ? y0 + 1 : H_low - 1;
    x0 = x0 > 0 ? x0 : 0;
    y0 = y0 > 0 ? y0 : 0;
    const f3 s00 = ldv(in_bloom, (size_t)y0 * W_low + x0), s10 = ldv(in_bloom, (size_t)y0 * W_low + x1);
    const f3 s01 = ldv(in_bloom, (size_t)y1 * W_low + x0), s11 = ldv(in_bloom, (size_t)y1 * W_low + x1);
    return lerp(lerp(s00, s10, u_frac), lerp(s01, s11, u_frac), v_frac);
}

// One pass of the chain with the sizes the reference's host loop hands to its kernels.
struct BloomPass {
    int kind;         // 0: down, bright pass fused (level 0); 1: down; 2: up-sample-add
    float *out;       // down: the level written; up: the image added to (row stride 2*b_w, as in the reference)
    const float *in;  // down: previous level / the frame; up: the lower level
    int a_w, a_h;     // down: in_W, in_H;  up: W_low, H_low
    int out_w, out_h; // pixels this pass writes: down in_W/2 x in_H/2, up 2*W_low x 2*H_low
};

PT_DEV void bloom_pass_pixel(const BloomPass &P, int x, int y) {
    if (P.kind == 2) {
        const size_t idx_high = (size_t)y * P.out_w + x;
        stv(P.out, idx_high, ldv(P.out, idx_high) + bloom_up_sample(P.in, P.a_w, P.a_h, x, y));
    } else {
        const f3 c = P.kind == 0 ? bloom_down_pixel<true>(P.in, P.a_w, P.a_h, x, y, 1.5f, 0.5f)
                                 : bloom_down_pixel<false>(P.in, P.a_w, P.a_h, x, y, 0.0f, 0.0f);
        stv(P.out, (size_t)y * P.out_w + x, c);
    }
}

__global__ __launch_bounds__(256) void bloom_pass_kernel(BloomPass P) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < P.out_w && y < P.out_h)
        bloom_pass_pixel(P, x, y);
}

// Up to 8 consecutive small passes in one workgroup; a workgroup lives on one CU and shares its L1,
// so the barrier's workgroup-scope release/acquire orders a pass's stores before the next pass's loads.
struct BloomSmallPasses {
    BloomPass p[8];
    int n;
};
__global__ __launch_bounds__(1024) void bloom_small_passes_kernel(BloomSmallPasses S) {
    for (int k = 0; k < S.n; ++k) {
        const BloomPass &P = S.p[k];
        const int total = P.out_w * P.out_h;
        for (int i = threadIdx.x; i < total; i += 1024)
            bloom_pass_pixel(P, i % P.out_w, i / P.out_w);
        __syncthreads();
    }
}

// Last up-sample-add (into the frame, which has EVEN width W = 2*W_low) + tonemap of every pixel.
// rgb8 may be NULL (an up-scale follows).
__global__ __launch_bounds__(256) void bloom_final_kernel(float *image, const float *in_bloom, int W, int H, int W_low,
                                                          int H_low, unsigned char *rgb8) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H)
        return;
    const size_t idx = (size_t)y * W + x;
    f3 c = ldv(image, idx);
    if (x < 2 * W_low && y < 2 * H_low) {
        c = c + bloom_up_sample(in_bloom, W_low, H_low, x, y);
        stv(image, idx, c);
    }
    if (rgb8) {
        unsigned char r, g, b;
        tonemap_pixel(c, r, g, b);
        const size_t o = ((size_t)(H - 1 - y) * W + x) * 3;
        rgb8[o] = r;
        rgb8[o + 1] = g;
        rgb8[o + 2] = b;
    }
}

// upscale_bilinear_kernel (+ tonemap of the up-scaled pixel; rgb8 rows flipped)
__global__ __launch_bounds__(256) void upscale_tonemap_kernel(float *out, const float *in, int out_w, int out_h, int in_w,
                                                              int in_h, unsigned char *rgb8) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out_w || y >= out_h)
        return;
    float u = ((float)x + 0.5f) * (float)in_w / (float)out_w - 0.5f;
    float v = ((float)y + 0.5f) * (float)in_h / (float)out_h - 0.5f;
    u = max_(0.0f, min_((float)(in_w - 1), u));
    v = max_(0.0f, min_((float)(in_h - 1), v));
    const int x0 = (int)__builtin_floorf(u), y0 = (int)__builtin_floorf(v);
    const int x1 = x0 + 1 < in_w - 1 ? x0 + 1 : in_w - 1, y1 = y0 + 1 < in_h - 1 ? y0 + 1 : in_h - 1;
    const float fx = u - (float)x0, fy = v - (float)y0;
    const f3 s00 = ldv(in, (size_t)y0 * in_w + x0), s10 = ldv(in, (size_t)y0 * in_w + x1);
    const f3 s01 = ldv(in, (size_t)y1 * in_w + x0), s11 = ldv(in, (size_t)y1 * in_w + x1);
    const f3 top = s00 * (1.0f - fx) + s10 * fx;
    const f3 bot = s01 * (1.0f - fx) + s11 * fx;
    const f3 result = top * (1.0f - fy) + bot * fy;
    stv(out, (size_t)y * out_w + x, result);
    unsigned char r, g, b;
    tonemap_pixel(result, r, g, b);
    const size_t o = ((size_t)(out_h - 1 - y) * out_w + x) * 3;
    rgb8[o] = r;
    rgb8[o + 1] = g;
    rgb8[o + 2] = b;
}

// stand-alone tonemap_kernel (only for frames whose width is odd, where the final up-sample-add
// cannot be fused: its row stride 2*floor(W/2) differs from W)
__global__ __launch_bounds__(256) void tonemap_only_kernel(unsigned char *rgb8, const float *in, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H)
        return;
    unsigned char r, g, b;
    tonemap_pixel(ldv(in, (size_t)y * W + x), r, g, b);
    const size_t o = ((size_t)(H - 1 - y) * W + x) * 3;
    rgb8[o] = r;
    rgb8[o + 1] = g;
    rgb8[o + 2] = b;
}

} // namespace pt
