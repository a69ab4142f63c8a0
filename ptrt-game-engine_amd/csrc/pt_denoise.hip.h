// pt_denoise.hip.h -- motion vectors + spatiotemporal denoiser + stand-alone tonemap (gfx950).
// the stage that follows the path tracer in every real preset.
//
// Reference (file:line of Mark-Rindler/PTRT-game-engine):
//   motion_vector_kernel          rendering/denoiser_kernels.cuh:33-68
//   firefly_suppression_kernel    rendering/denoiser.cuh:376-424
//   temporal_accumulation_kernel  rendering/denoiser.cuh:426-584 (edge-aware taps 231-374)
//   estimate_variance_kernel      rendering/denoiser.cuh:586-648
//   atrous_filter_kernel          rendering/denoiser.cuh:650-749
//   Denoiser::denoiseChannel      rendering/denoiser.cuh:884-964 (non-split path)
//   tonemap_kernel                scene/scene.cuh:2004-2047
//
// These are HBM-bound image passes.  What is done differently from the reference's launch list:
//   * no device-to-device copies: the six cudaMemcpy per channel per frame (history <- result,
//     denoiser.cuh:922-930,1049-1061) become pointer swaps of double-buffered history sets;
//   * init_moments + the first-frame history copy are folded into the temporal kernel's
//     first-frame branch; the a-trous chain reads the accumulated mean in place and ping-pongs
//     two scratch images;
//   * 64x4-pixel workgroups so a wave touches one contiguous row segment.
// The reference runs temporal_accumulation in place (out_mean == current_color, a data race);
// here every read sees the pre-kernel image (see oracle/denoiser_oracle.cpp).
#pragma once
#include "pt_device.hip.h"

namespace pt {

struct DenoiseSettings { // DenoiserSettings (denoiser.cuh:36-73), non-split subset
    float tau, min_alpha, max_history, sigma_luminance, sigma_normal, sigma_depth;
    int atrous_iterations;
    float clamp_scale, firefly_threshold;
    float depth_reject_absolute, depth_reject_relative, normal_reject_threshold, sky_depth_threshold;
    float edge_depth_threshold, edge_normal_threshold;
    int use_object_ids, enable_firefly_suppression;
};

PT_DEV f3 ld3(const float *p, size_t i) { return mk3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); }
PT_DEV void st3(float *p, size_t i, f3 v) {
    p[i * 3] = v.x;
    p[i * 3 + 1] = v.y;
    p[i * 3 + 2] = v.z;
}
PT_DEV f3 max3(f3 a, f3 b) { return mk3(max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z)); }
PT_DEV f3 min3(f3 a, f3 b) { return mk3(min_(a.x, b.x), min_(a.y, b.y), min_(a.z, b.z)); }
PT_DEV float luminance(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
PT_DEV int clampi(int v, int a, int b) { return v < a ? a : (v > b ? b : v); }
PT_DEV bool is_sky(float depth, f3 n, float thr) { return (depth > thr) || (dot(n, n) < 0.1f); }
PT_DEV bool edge_disc(float d0, float d1, f3 n0, f3 n1, int o0, int o1, float dthr, float nthr, bool use_obj) {
    if (use_obj && o0 != o1 && o0 >= 0 && o1 >= 0)
        return true;
    const float max_d = max_(d0, d1);
    const float dd = __builtin_fabsf(d0 - d1);
    if (max_d > 1e-6f && dd / max_d > dthr)
        return true;
    return dot(n0, n1) < nthr;
}

#define PT_PIXEL_XY                                                                                            \
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);                                                        \
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);                                                         \
    if (x >= W || y >= H)                                                                                      \
        return;                                                                                                \
    const int idx = y * W + x;

// ------------------------------------------------------------------ motion vectors
__global__ __launch_bounds__(256) void motion_vector_kernel(float *__restrict__ out_mv, const float *__restrict__ depth,
                                                            int W, int H, f3 origin, f3 llc, f3 hor, f3 ver,
                                                            const float *__restrict__ pvp) {
    PT_PIXEL_XY
    const float d = depth[idx];
    if (d >= 1e29f) {
        out_mv[idx * 2] = 0.0f;
        out_mv[idx * 2 + 1] = 0.0f;
        return;
    }
    const float u = ((float)x + 0.5f) / (float)W;
    const float v = ((float)y + 0.5f) / (float)H;
    const float s = u, t = 1.0f - v;
    const f3 rd = llc + hor * s + ver * t - origin;
    const f3 dir = normalize(rd);
    const f3 wp = origin + dir * d;
    const float cx = pvp[0] * wp.x + pvp[4] * wp.y + pvp[8] * wp.z + pvp[12] * 1.0f;
    const float cy = pvp[1] * wp.x + pvp[5] * wp.y + pvp[9] * wp.z + pvp[13] * 1.0f;
    const float cw = pvp[3] * wp.x + pvp[7] * wp.y + pvp[11] * wp.z + pvp[15] * 1.0f;
    const float ndc_x = cx / cw, ndc_y = cy / cw;
    const float prev_u = (ndc_x + 1.0f) * 0.5f;
    const float prev_v = (1.0f - ndc_y) * 0.5f;
    out_mv[idx * 2] = u - prev_u;
    out_mv[idx * 2 + 1] = v - prev_v;
}

// ------------------------------------------------------------------ firefly suppression
__global__ __launch_bounds__(256) void firefly_kernel(float *__restrict__ out, const float *__restrict__ in,
                                                      const float *__restrict__ depth, const float *__restrict__ normal,
                                                      float sky, int W, int H, int enabled) {
    PT_PIXEL_XY
    const f3 center = ld3(in, idx);
    if (!enabled || is_sky(depth[idx], ld3(normal, idx), sky)) {
        st3(out, idx, center);
        return;
    }
    f3 mx = mk3(0.0f);
    bool any = false;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0)
                continue;
            const int nx = x + dx, ny = y + dy;
            if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
                mx = max3(mx, ld3(in, ny * W + nx));
                any = true;
            }
        }
    if (any) {
        f3 c = min3(center, mx * 1.25f);
        c = min3(c, mk3(10.0f));
        st3(out, idx, c);
    } else {
        st3(out, idx, center);
    }
}

// ------------------------------------------------------------------ temporal accumulation
struct Taps {
    int idx[4];
    float w[4];
    bool valid[4];
    float total_w;
    int nearest;
};
PT_DEV f3 sample3(const float *buf, const Taps &t) {
    if (t.total_w < 1e-6f) {
        for (int k = 0; k < 4; ++k)
            if (t.valid[k])
                return ld3(buf, t.idx[k]);
        return ld3(buf, t.nearest);
    }
    return (ld3(buf, t.idx[0]) * t.w[0] + ld3(buf, t.idx[1]) * t.w[1] + ld3(buf, t.idx[2]) * t.w[2] +
            ld3(buf, t.idx[3]) * t.w[3]) *
           (1.0f / t.total_w);
}
PT_DEV float sample1(const float *buf, const Taps &t) {
    if (t.total_w < 1e-6f) {
        for (int k = 0; k < 4; ++k)
            if (t.valid[k])
                return buf[t.idx[k]];
        return buf[t.nearest];
    }
    return (buf[t.idx[0]] * t.w[0] + buf[t.idx[1]] * t.w[1] + buf[t.idx[2]] * t.w[2] + buf[t.idx[3]] * t.w[3]) *
           (1.0f / t.total_w);
}

// cur: firefly-filtered image.  prev_* : last frame's history (on the first frame they are not
// read: the history IS the current image, init_moments folded in).  out_*: the new history.
__global__ __launch_bounds__(256) void temporal_kernel(
    float *__restrict__ out_mean, float *__restrict__ out_m2, float *__restrict__ out_len, const float *__restrict__ cur,
    const float *__restrict__ prev_mean, const float *__restrict__ prev_m2, const float *__restrict__ prev_len,
    const float *__restrict__ motion, const float *__restrict__ depth, const float *__restrict__ prev_depth,
    const float *__restrict__ normal, const float *__restrict__ prev_normal, const int *__restrict__ object_id,
    const int *__restrict__ prev_object_id, DenoiseSettings S, int first_frame, int W, int H) {
    PT_PIXEL_XY
    const bool use_obj = S.use_object_ids != 0;
    const f3 cur_c = ld3(cur, idx);
    const float d = depth[idx];
    const f3 n = ld3(normal, idx);
    const int obj_id = use_obj ? object_id[idx] : -1;
    if (is_sky(d, n, S.sky_depth_threshold)) {
        st3(out_mean, idx, cur_c);
        st3(out_m2, idx, cur_c * cur_c);
        out_len[idx] = 1.0f;
        return;
    }
    f3 nmean = mk3(0.0f), nm2 = mk3(0.0f);
    int ncount = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int ni = clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1);
            const f3 nc = ld3(cur, ni);
            const int no = use_obj ? object_id[ni] : -1;
            if (!edge_disc(d, depth[ni], n, ld3(normal, ni), obj_id, no, S.edge_depth_threshold, S.edge_normal_threshold,
                           use_obj)) {
                nmean = nmean + nc;
                nm2 = nm2 + nc * nc;
                ncount++;
            }
        }
    if (ncount == 0) {
        nmean = cur_c;
        nm2 = cur_c * cur_c;
        ncount = 1;
    }
    const float inv_n = 1.0f / (float)ncount;
    nmean = nmean * inv_n;
    nm2 = nm2 * inv_n;
    const f3 nvar = max3(nm2 - nmean * nmean, mk3(0.0f));
    const f3 nstd = mk3(__builtin_sqrtf(nvar.x), __builtin_sqrtf(nvar.y), __builtin_sqrtf(nvar.z));
    const f3 soft_min = nmean - nstd * S.clamp_scale;
    const f3 soft_max = nmean + nstd * S.clamp_scale;

    const float prev_u = (float)x + 0.5f - motion[idx * 2] * (float)W;
    const float prev_v = (float)y + 0.5f - motion[idx * 2 + 1] * (float)H;
    bool valid = !(prev_u < 0.5f || prev_v < 0.5f || prev_u >= (float)((float)W - 0.5f) || prev_v >= (float)((float)H - 0.5f));
    f3 hist_mean = mk3(0.0f), hist_m2 = mk3(0.0f);
    float hist_len = 0.0f;
    if (valid) {
        // on the first frame the history is this frame: mean = cur, m2 = cur^2, len = 1, G-buffers = current
        const float *pm = first_frame ? cur : prev_mean;
        const float *pd = first_frame ? depth : prev_depth;
        const float *pn = first_frame ? normal : prev_normal;
        const int *po = first_frame ? object_id : prev_object_id;
        const float fx = prev_u - 0.5f, fy = prev_v - 0.5f;
        int x0 = (int)__builtin_floorf(fx), y0 = (int)__builtin_floorf(fy);
        int x1 = x0 + 1, y1 = y0 + 1;
        const float sx = fx - (float)x0, sy = fy - (float)y0;
        x0 = clampi(x0, 0, W - 1);
        y0 = clampi(y0, 0, H - 1);
        x1 = clampi(x1, 0, W - 1);
        y1 = clampi(y1, 0, H - 1);
        Taps t;
        t.idx[0] = y0 * W + x0;
        t.idx[1] = y0 * W + x1;
        t.idx[2] = y1 * W + x0;
        t.idx[3] = y1 * W + x1;
        const float bw[4] = {(1.0f - sx) * (1.0f - sy), sx * (1.0f - sy), (1.0f - sx) * sy, sx * sy};
        for (int k = 0; k < 4; ++k) {
            const int o = use_obj ? po[t.idx[k]] : -1;
            t.valid[k] = !edge_disc(d, pd[t.idx[k]], n, ld3(pn, t.idx[k]), obj_id, o, S.edge_depth_threshold,
                                    S.edge_normal_threshold, use_obj);
            t.w[k] = t.valid[k] ? bw[k] : 0.0f;
        }
        t.total_w = t.w[0] + t.w[1] + t.w[2] + t.w[3];
        t.nearest = clampi((int)__builtin_floorf(prev_v), 0, H - 1) * W + clampi((int)__builtin_floorf(prev_u), 0, W - 1);
        hist_mean = sample3(pm, t);
        if (first_frame) {
            // m2 history = cur*cur per texel, len history = 1: sample those images through the same taps
            if (t.total_w < 1e-6f) {
                int pick = t.nearest;
                for (int k = 3; k >= 0; --k)
                    if (t.valid[k])
                        pick = t.idx[k];
                const f3 c = ld3(cur, pick);
                hist_m2 = c * c;
                hist_len = 1.0f;
            } else {
                const f3 c0 = ld3(cur, t.idx[0]), c1 = ld3(cur, t.idx[1]), c2 = ld3(cur, t.idx[2]), c3 = ld3(cur, t.idx[3]);
                hist_m2 = ((c0 * c0) * t.w[0] + (c1 * c1) * t.w[1] + (c2 * c2) * t.w[2] + (c3 * c3) * t.w[3]) * (1.0f / t.total_w);
                hist_len = (1.0f * t.w[0] + 1.0f * t.w[1] + 1.0f * t.w[2] + 1.0f * t.w[3]) * (1.0f / t.total_w);
            }
        } else {
            hist_m2 = sample3(prev_m2, t);
            hist_len = sample1(prev_len, t);
        }
        const float hist_d = sample1(pd, t);
        if (use_obj && po[t.nearest] != obj_id)
            valid = false;
        const float dad = __builtin_fabsf(d - hist_d);
        if (dad > S.depth_reject_absolute || dad > S.depth_reject_relative * max_(1e-6f, d))
            valid = false;
        if (dot(n, ld3(pn, t.nearest)) < S.normal_reject_threshold)
            valid = false;
    }
    if (valid)
        hist_mean = min3(max3(hist_mean, soft_min), soft_max);
    float alpha = 1.0f, nlen = 1.0f;
    if (valid) {
        const f3 var = max3(hist_m2 - (hist_mean * hist_mean), mk3(0.0f));
        const float std_approx = (__builtin_sqrtf(var.x) + __builtin_sqrtf(var.y) + __builtin_sqrtf(var.z)) * (1.0f / 3.0f);
        const float variance_alpha = std_approx / (std_approx + S.tau);
        const float history_alpha = 1.0f / (hist_len + 1.0f);
        alpha = clampf(max_(variance_alpha, history_alpha), S.min_alpha, 1.0f);
        nlen = min_(hist_len + 1.0f, S.max_history);
    }
    st3(out_mean, idx, hist_mean * (1.0f - alpha) + cur_c * alpha);
    st3(out_m2, idx, hist_m2 * (1.0f - alpha) + (cur_c * cur_c) * alpha);
    out_len[idx] = nlen;
}

// ------------------------------------------------------------------ variance estimate
__global__ __launch_bounds__(256) void variance_kernel(float *__restrict__ out_var, const float *__restrict__ color,
                                                       const float *__restrict__ m2, const float *__restrict__ hlen,
                                                       const float *__restrict__ depth, const float *__restrict__ normal,
                                                       const int *__restrict__ object_id, float sky, int use_obj_i, int W,
                                                       int H) {
    PT_PIXEL_XY
    const bool uo = use_obj_i != 0;
    const float d = depth[idx];
    const f3 n = ld3(normal, idx);
    const int obj = uo ? object_id[idx] : -1;
    if (is_sky(d, n, sky)) {
        out_var[idx] = 0.0f;
        return;
    }
    const f3 c = ld3(color, idx), cm2 = ld3(m2, idx);
    const f3 var = max3(cm2 - (c * c), mk3(0.0f));
    const float reliability = min_(hlen[idx] * 0.25f, 1.0f);
    const float boost = 1.0f + (1.0f - reliability) * 3.0f;
    f3 smean = mk3(0.0f), sm2 = mk3(0.0f);
    int count = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int ni = clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1);
            if (uo && object_id[ni] != obj)
                continue;
            const f3 nc = ld3(color, ni);
            smean = smean + nc;
            sm2 = sm2 + nc * nc;
            count++;
        }
    const float inv = 1.0f / (float)count;
    smean = smean * inv;
    sm2 = sm2 * inv;
    const f3 svar = max3(sm2 - smean * smean, mk3(0.0f));
    const f3 cv = max3(var * boost, svar);
    out_var[idx] = 0.2126f * cv.x + 0.7152f * cv.y + 0.0722f * cv.z;
}

// ------------------------------------------------------------------ a-trous wavelet pass
__global__ __launch_bounds__(256) void atrous_kernel(float *__restrict__ out, float *__restrict__ out_var,
                                                     const float *__restrict__ in, const float *__restrict__ in_var,
                                                     const float *__restrict__ normal, const float *__restrict__ depth,
                                                     const int *__restrict__ object_id, int step, float sigma_lum,
                                                     float sky, float edt, float ent, int use_obj_i, int W, int H) {
    PT_PIXEL_XY
    constexpr float KW[5] = {1.0f, 4.0f, 6.0f, 4.0f, 1.0f};
    const bool uo = use_obj_i != 0;
    const f3 cc = ld3(in, idx), cn = ld3(normal, idx);
    const float cd = depth[idx];
    const int cobj = uo ? object_id[idx] : -1;
    const float cvar = in_var[idx];
    const float clum = luminance(cc);
    if (is_sky(cd, cn, sky)) {
        st3(out, idx, cc);
        out_var[idx] = cvar;
        return;
    }
    const float var_scale = __builtin_sqrtf(max_(cvar, 1e-6f));
    const float asl = sigma_lum * (1.0f + var_scale * 2.0f);
    const float inv_sl2 = 1.0f / (2.0f * asl * asl + 1e-6f);
    f3 sum = mk3(0.0f);
    float sum_var = 0.0f, total_w = 0.0f;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const int nx = x + dx * step, ny = y + dy * step;
            if (nx < 0 || nx >= W || ny < 0 || ny >= H)
                continue;
            const int ni = ny * W + nx;
            if (uo) {
                const int nobj = object_id[ni];
                if (cobj != nobj && cobj >= 0 && nobj >= 0)
                    continue;
            }
            const float nd = depth[ni];
            const float max_d = max_(cd, nd);
            const float dd = __builtin_fabsf(cd - nd);
            if (max_d > 1e-6f && dd / max_d > edt)
                continue;
            const f3 nn = ld3(normal, ni);
            if (dot(cn, nn) < ent)
                continue;
            if (is_sky(nd, nn, sky))
                continue;
            const f3 nc = ld3(in, ni);
            const float ld = __builtin_fabsf(clum - luminance(nc));
            const float wl = det_exp(-ld * ld * inv_sl2);
            // atrous_kernel[k] = (a*b)/256 with a,b in {1,4,6}: products and the /256 are exact in fp32
            const float weight = (KW[dy + 2] * KW[dx + 2] / 256.0f) * wl;
            sum = sum + nc * weight;
            sum_var += in_var[ni] * weight;
            total_w += weight;
        }
    if (total_w < 1e-6f) {
        st3(out, idx, cc);
        out_var[idx] = cvar;
    } else {
        const float inv_w = 1.0f / total_w;
        st3(out, idx, sum * inv_w);
        out_var[idx] = sum_var * inv_w;
    }
}

// ------------------------------------------------------------------ tonemap_kernel (scene.cuh:2004-2047)
__global__ __launch_bounds__(256) void tonemap_kernel(unsigned char *__restrict__ out, const float *__restrict__ in, int W,
                                                      int H) {
    PT_PIXEL_XY
    unsigned char r, g, b;
    tonemap_pixel(ld3(in, idx), r, g, b);
    const size_t o = ((size_t)(H - 1 - y) * W + x) * 3;
    out[o] = r;
    out[o + 1] = g;
    out[o + 2] = b;
}

} // namespace pt
