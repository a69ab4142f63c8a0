// pt_denoise.hip.h -- motion vectors + spatiotemporal denoiser + tonemap (gfx950).
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
// These are image passes whose first cut (one dword load per component, as the reference's
// vec3/float arrays suggest) was bound by vector-memory INSTRUCTION issue, not bandwidth: an
// a-trous tap cost 9 dword loads, 225 per pixel, and a pass ran at 7 % of the HBM roofline.
// Layout here (all internal; the API-visible vec3/float buffers are untouched):
//     G4  = {normal.xyz, depth}      one global_load_dwordx4 per tap, double-buffered so the
//                                    previous frame's G-buffer needs no copy
//     C4  = {rgb, variance}          a-trous ping-pong images
//     H1  = {mean.rgb, history len}, H2 = {m2.rgb, -}   history, double-buffered (swap, no copies)
// so a tap is 2 x dwordx4 + 1 dword.  The reference's six cudaMemcpy per channel per frame
// (denoiser.cuh:922-930,1049-1061) become pointer swaps; init_moments + the first-frame history
// copy are a branch of the temporal kernel; the last a-trous pass writes the API's vec3 image and
// the tonemapped RGB8 in the same store phase.  Arithmetic per pixel is the reference's, term by
// term (same contract as the path: dot() fused, __expf -> det_exp), so results are bit-identical
// to oracle/denoiser_oracle.cpp.  The reference runs temporal_accumulation in place (a race);
// here every read sees the pre-kernel image, as in the oracle.
#pragma once
#include <type_traits>
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
PT_DEV f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
PT_DEV float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
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

// ------------------------------------------------------------------ prep: G4 pack + motion vectors + firefly
// One pass over the path tracer's outputs: packs {normal, depth}, writes the motion vector
// (when enabled) and the firefly-suppressed colour image {rgb, 0}.
__global__ __launch_bounds__(256) void prep_kernel(float4 *__restrict__ g4, float *__restrict__ out_mv,
                                                   float4 *__restrict__ cur4, const float *__restrict__ accum,
                                                   const float *__restrict__ normal, const float *__restrict__ depth, int W,
                                                   int H, f3 origin, f3 llc, f3 hor, f3 ver, f3 cam_u, f3 cam_v,
                                                   float lens_radius, const float *__restrict__ pvp, int do_motion,
                                                   float sky, int firefly) {
    PT_PIXEL_XY
    const float d = depth[idx];
    const f3 n = ld3(normal, idx);
    g4[idx] = mk4(n, d);
    if (do_motion) { // motion_vector_kernel
        if (d >= 1e29f) {
            out_mv[idx * 2] = 0.0f;
            out_mv[idx * 2 + 1] = 0.0f;
        } else {
            const float u = ((float)x + 0.5f) / (float)W;
            const float v = ((float)y + 0.5f) / (float)H;
            const float s = u, t = 1.0f - v;
            f3 ray_o = origin;
            f3 rd = llc + hor * s + ver * t - origin;
            if (lens_radius > 0) { // Camera::get_ray, device branch (camera.cuh:177-185): lens sample hashed from (s, t)
                const uint32_t hx = (uint32_t)(s * 10000.0f) + (uint32_t)(t * 5000.0f);
                const uint32_t hy = (uint32_t)(t * 10000.0f) + (uint32_t)(s * 5000.0f);
                uint32_t seed = (hx * 1973u) ^ (hy * 9277u) ^ 0x9e3779b9u; // random_in_unit_disk_hash (camera.cuh:55-72)
                seed ^= seed >> 17;
                seed *= 0xed5ad4bbu;
                seed ^= seed >> 11;
                seed *= 0xac4c1b51u;
                seed ^= seed >> 15;
                seed *= 0x31848babu;
                seed ^= seed >> 14;
                const float r1 = ((float)(seed & 0xFFFFu) + 0.5f) / 65536.0f;
                const float r2 = ((float)((seed * 0x343fdu + 0xc0f5u) & 0xFFFFu) + 0.5f) / 65536.0f;
                const float r = sqrt_ieee(r1);
                const float phi = 6.2831853f * r2;
                const f3 disk = mk3(r * det_cos(phi), r * det_sin(phi), 0.0f) * lens_radius;
                const f3 offset = cam_u * disk.x + cam_v * disk.y;
                rd = llc + hor * s + ver * t - origin - offset;
                ray_o = origin + offset;
            }
            const f3 dir = normalize(rd);
            const f3 wp = ray_o + dir * d;
            const float cx = pvp[0] * wp.x + pvp[4] * wp.y + pvp[8] * wp.z + pvp[12] * 1.0f;
            const float cy = pvp[1] * wp.x + pvp[5] * wp.y + pvp[9] * wp.z + pvp[13] * 1.0f;
            const float cw = pvp[3] * wp.x + pvp[7] * wp.y + pvp[11] * wp.z + pvp[15] * 1.0f;
            const float ndc_x = cx / cw, ndc_y = cy / cw;
            const float prev_u = (ndc_x + 1.0f) * 0.5f;
            const float prev_v = (1.0f - ndc_y) * 0.5f;
            out_mv[idx * 2] = u - prev_u;
            out_mv[idx * 2 + 1] = v - prev_v;
        }
    }
    // firefly_suppression_kernel
    const f3 center = ld3(accum, idx);
    f3 outc = center;
    if (firefly && !is_sky(d, n, sky)) {
        f3 mx = mk3(0.0f);
        bool any = false;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (dx == 0 && dy == 0)
                    continue;
                const int nx = x + dx, ny = y + dy;
                if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
                    mx = max3(mx, ld3(accum, ny * W + nx));
                    any = true;
                }
            }
        if (any) {
            outc = min3(center, mx * 1.25f);
            outc = min3(outc, mk3(10.0f));
        }
    }
    cur4[idx] = mk4(outc, 0.0f);
}

// ------------------------------------------------------------------ temporal accumulation
struct Taps {
    int idx[4];
    float w[4];
    bool valid[4];
    float total_w;
    int nearest;
};
PT_DEV int first_valid(const Taps &t) {
    int pick = t.nearest;
    for (int k = 3; k >= 0; --k)
        if (t.valid[k])
            pick = t.idx[k];
    return pick;
}
PT_DEV f3 blend3(f3 a, f3 b, f3 c, f3 d, const Taps &t) {
    return (a * t.w[0] + b * t.w[1] + c * t.w[2] + d * t.w[3]) * (1.0f / t.total_w);
}
PT_DEV float blend1(float a, float b, float c, float d, const Taps &t) {
    return (a * t.w[0] + b * t.w[1] + c * t.w[2] + d * t.w[3]) * (1.0f / t.total_w);
}

// cur4: firefly-filtered image.  ph1/ph2/pg4/pobj: last frame's history and G-buffer (on the first
// frame the history IS the current frame: mean = cur, m2 = cur^2, len = 1, G = current).
// (Round 3, measured and dropped: every load of this kernel -- the 3x3 neighbourhood's colours, the four taps' G-buffer
// entries, ids and histories, the nearest texel -- issued up front and unconditionally, since every address is known before any
// test: 112-114 us against 103-116 as written, at 94 instead of 72 VGPRs.  Its 80 % of wave-cycles in s_waitcnt are not the
// dependent chains of the source.)
// (Round 4: the 3x3 neighbourhood -- colour, {normal, depth}, object id of nine pixels, 27 gathers per thread through the
// texture addresser -- comes out of an LDS copy of the workgroup's 66 x 6 footprint, staged with coalesced loads as the a-trous
// passes do; the four history taps sit at motion-dependent addresses and stay gathers.)
constexpr int TP_SPAN = 64 + 2, TP_ENT = TP_SPAN * (4 + 2);
__global__ __launch_bounds__(256) void temporal_kernel(float4 *__restrict__ oh1, float4 *__restrict__ oh2,
                                                       const float4 *__restrict__ cur4, const float4 *__restrict__ ph1,
                                                       const float4 *__restrict__ ph2, const float *__restrict__ motion,
                                                       const float4 *__restrict__ g4, const float4 *__restrict__ pg4,
                                                       const int *__restrict__ object_id, const int *__restrict__ pobj,
                                                       DenoiseSettings S, int first_frame, int W, int H) {
    __shared__ float4 s_c[TP_ENT], s_g[TP_ENT];
    __shared__ int s_o[TP_ENT];
    const bool use_obj = S.use_object_ids != 0;
    {
        const int bx = blockIdx.x * 64 - 1, by = blockIdx.y * 4 - 1;
        for (int e = threadIdx.x; e < TP_ENT; e += 256) { // entry (ex, ey) = pixel (clamp(bx + ex), clamp(by + ey)): the taps' own clamping
            const int ey = e / TP_SPAN, ex = e - ey * TP_SPAN;
            const int pi = clampi(by + ey, 0, H - 1) * W + clampi(bx + ex, 0, W - 1);
            s_c[e] = cur4[pi];
            s_g[e] = g4[pi];
            s_o[e] = use_obj ? object_id[pi] : -1;
        }
    }
    __syncthreads();
    PT_PIXEL_XY
    const int ce = ((threadIdx.x >> 6) + 1) * TP_SPAN + (threadIdx.x & 63) + 1;
    const f3 cur_c = xyz(s_c[ce]);
    const float4 g = s_g[ce];
    const float d = g.w;
    const f3 n = xyz(g);
    const int obj_id = s_o[ce];
    if (is_sky(d, n, S.sky_depth_threshold)) {
        oh1[idx] = mk4(cur_c, 1.0f);
        oh2[idx] = mk4(cur_c * cur_c, 0.0f);
        return;
    }
    f3 nmean = mk3(0.0f), nm2 = mk3(0.0f);
    int ncount = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ne = ce + dy * TP_SPAN + dx;
            const float4 ng = s_g[ne];
            const int no = s_o[ne];
            if (!edge_disc(d, ng.w, n, xyz(ng), obj_id, no, S.edge_depth_threshold, S.edge_normal_threshold, use_obj)) {
                const f3 nc = xyz(s_c[ne]);
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
    const f3 nstd = mk3(sqrt_ieee(nvar.x), sqrt_ieee(nvar.y), sqrt_ieee(nvar.z));
    const f3 soft_min = nmean - nstd * S.clamp_scale;
    const f3 soft_max = nmean + nstd * S.clamp_scale;

    const float prev_u = (float)x + 0.5f - motion[idx * 2] * (float)W;
    const float prev_v = (float)y + 0.5f - motion[idx * 2 + 1] * (float)H;
    bool valid = !(prev_u < 0.5f || prev_v < 0.5f || prev_u >= (float)((float)W - 0.5f) || prev_v >= (float)((float)H - 0.5f));
    f3 hist_mean = mk3(0.0f), hist_m2 = mk3(0.0f);
    float hist_len = 0.0f;
    if (valid) {
        const float4 *hg = first_frame ? g4 : pg4;
        const int *po = first_frame ? object_id : pobj;
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
        float4 tg[4];
        for (int k = 0; k < 4; ++k) {
            tg[k] = hg[t.idx[k]];
            const int o = use_obj ? po[t.idx[k]] : -1;
            t.valid[k] = !edge_disc(d, tg[k].w, n, xyz(tg[k]), obj_id, o, S.edge_depth_threshold, S.edge_normal_threshold,
                                    use_obj);
            t.w[k] = t.valid[k] ? bw[k] : 0.0f;
        }
        t.total_w = t.w[0] + t.w[1] + t.w[2] + t.w[3];
        t.nearest = clampi((int)__builtin_floorf(prev_v), 0, H - 1) * W + clampi((int)__builtin_floorf(prev_u), 0, W - 1);
        float hist_d;
        if (t.total_w < 1e-6f) { // no usable tap: first valid one, else the nearest texel (denoiser.cuh:294-306)
            const int pick = first_valid(t);
            if (first_frame) {
                const f3 c = xyz(cur4[pick]);
                hist_mean = c;
                hist_m2 = c * c;
                hist_len = 1.0f;
            } else {
                const float4 a = ph1[pick];
                hist_mean = xyz(a);
                hist_len = a.w;
                hist_m2 = xyz(ph2[pick]);
            }
            hist_d = hg[pick].w;
        } else {
            if (first_frame) {
                const f3 c0 = xyz(cur4[t.idx[0]]), c1 = xyz(cur4[t.idx[1]]), c2 = xyz(cur4[t.idx[2]]), c3 = xyz(cur4[t.idx[3]]);
                hist_mean = blend3(c0, c1, c2, c3, t);
                hist_m2 = blend3(c0 * c0, c1 * c1, c2 * c2, c3 * c3, t);
                hist_len = blend1(1.0f, 1.0f, 1.0f, 1.0f, t);
            } else {
                const float4 a0 = ph1[t.idx[0]], a1 = ph1[t.idx[1]], a2 = ph1[t.idx[2]], a3 = ph1[t.idx[3]];
                hist_mean = blend3(xyz(a0), xyz(a1), xyz(a2), xyz(a3), t);
                hist_len = blend1(a0.w, a1.w, a2.w, a3.w, t);
                hist_m2 = blend3(xyz(ph2[t.idx[0]]), xyz(ph2[t.idx[1]]), xyz(ph2[t.idx[2]]), xyz(ph2[t.idx[3]]), t);
            }
            hist_d = blend1(tg[0].w, tg[1].w, tg[2].w, tg[3].w, t);
        }
        if (use_obj && po[t.nearest] != obj_id)
            valid = false;
        const float dad = __builtin_fabsf(d - hist_d);
        if (dad > S.depth_reject_absolute || dad > S.depth_reject_relative * max_(1e-6f, d))
            valid = false;
        if (dot(n, xyz(hg[t.nearest])) < S.normal_reject_threshold)
            valid = false;
    }
    if (valid)
        hist_mean = min3(max3(hist_mean, soft_min), soft_max);
    float alpha = 1.0f, nlen = 1.0f;
    if (valid) {
        const f3 var = max3(hist_m2 - (hist_mean * hist_mean), mk3(0.0f));
        const float std_approx = (sqrt_ieee(var.x) + sqrt_ieee(var.y) + sqrt_ieee(var.z)) * (1.0f / 3.0f);
        const float variance_alpha = std_approx / (std_approx + S.tau);
        const float history_alpha = 1.0f / (hist_len + 1.0f);
        alpha = clampf(max_(variance_alpha, history_alpha), S.min_alpha, 1.0f);
        nlen = min_(hist_len + 1.0f, S.max_history);
    }
    oh1[idx] = mk4(hist_mean * (1.0f - alpha) + cur_c * alpha, nlen);
    oh2[idx] = mk4(hist_m2 * (1.0f - alpha) + (cur_c * cur_c) * alpha, 0.0f);
}

// ------------------------------------------------------------------ variance estimate -> C4 = {mean, variance}
__global__ __launch_bounds__(256) void variance_kernel(float4 *__restrict__ out_c4, const float4 *__restrict__ h1,
                                                       const float4 *__restrict__ h2, const float4 *__restrict__ g4,
                                                       const int *__restrict__ object_id, int *__restrict__ hist_obj,
                                                       float sky, int use_obj_i, int W, int H) {
    PT_PIXEL_XY
    const bool uo = use_obj_i != 0;
    const float4 g = g4[idx];
    const float4 a = h1[idx];
    const f3 c = xyz(a);
    const int cur_obj = object_id[idx];
    hist_obj[idx] = cur_obj; // next frame's history object ids (the temporal pass of this frame is done)
    const int obj = uo ? cur_obj : -1;
    if (is_sky(g.w, xyz(g), sky)) {
        out_c4[idx] = mk4(c, 0.0f);
        return;
    }
    const f3 cm2 = xyz(h2[idx]);
    const f3 var = max3(cm2 - (c * c), mk3(0.0f));
    const float reliability = min_(a.w * 0.25f, 1.0f);
    const float boost = 1.0f + (1.0f - reliability) * 3.0f;
    f3 smean = mk3(0.0f), sm2 = mk3(0.0f);
    int count = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int ni = clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1);
            if (uo && object_id[ni] != obj)
                continue;
            const f3 nc = xyz(h1[ni]);
            smean = smean + nc;
            sm2 = sm2 + nc * nc;
            count++;
        }
    const float inv = 1.0f / (float)count;
    smean = smean * inv;
    sm2 = sm2 * inv;
    const f3 svar = max3(sm2 - smean * smean, mk3(0.0f));
    const f3 cv = max3(var * boost, svar);
    out_c4[idx] = mk4(c, 0.2126f * cv.x + 0.7152f * cv.y + 0.0722f * cv.z);
}

// ------------------------------------------------------------------ a-trous wavelet pass on C4
// A pass with step s combines a pixel only with pixels of its own row class (y mod s), so a
// workgroup takes 64 CONSECUTIVE pixels of 4 rows of one class: the footprint of its 25 taps is
// 8 rows (the class's neighbours) x (64 + 4s) consecutive pixels, staged in LDS once with
// coalesced dwordx4 loads (<= 36 KB at s = 16: {rgb,var} + {normal,depth} + object id per entry)
// and read back with ds_read_b128, instead of being gathered 25 times through a 32-KB L1 that
// eight resident workgroups thrash (first cut: 74 % of wave-cycles in s_waitcnt, FETCH_SIZE 2.9x
// the image).  Tiling BOTH axes by class (16x16 pixels of one (x mod s, y mod s)) staged less
// but made every global access a lone 16-B piece of its sector: faster for s <= 4, slower for
// s >= 8 (133 / 156 us) -- rows by class, columns contiguous is coalesced for every s.
// LAST: also writes the API's vec3 image and the tonemapped RGB8 (rows flipped), saving two passes.
constexpr int AT_W = 64, AT_ROWS = 4;
constexpr int AT_OUTSIDE = (int)0x80000000; // object-id slot of a footprint entry that lies outside the image
PT_DEV int atrous_span(int step) { return AT_W + 4 * step; }
inline size_t atrous_lds_bytes(int step) { return (size_t)(AT_W + 4 * step) * (AT_ROWS + 4) * 40; }
// FAST (option "atrous_exp" 1): the luminance weight through the hardware exponential, v_exp_f32(x log2 e) -- what the reference
// itself computes there (`__expf`, denoiser.cuh:731) -- instead of the 27-instruction deterministic one the oracle defines;
// a stated-tolerance mode (tests/test_denoiser.py), the bit-exact mode is the default.
template <bool LAST, bool FAST = false>
__global__ __launch_bounds__(256) void atrous_kernel(float4 *__restrict__ out_c4, const float4 *__restrict__ in_c4,
                                                     const float4 *__restrict__ g4, const int *__restrict__ object_id,
                                                     int step, float sigma_lum, float sky, float edt, float ent, int use_obj_i,
                                                     int W, int H, float *__restrict__ out3, unsigned char *__restrict__ rgb8) {
    extern __shared__ float4 at_lds[];
    constexpr float KW[5] = {1.0f, 4.0f, 6.0f, 4.0f, 1.0f};
    const bool uo = use_obj_i != 0;
    const int span = atrous_span(step), entries = span * (AT_ROWS + 4);
    float4 *s_c = at_lds, *s_g = at_lds + entries;
    int *s_o = reinterpret_cast<int *>(at_lds + 2 * entries);
    float *s_l = reinterpret_cast<float *>(s_o + entries); // luminance of the entry's colour: computed once here, not by each of the 25 taps that read it
    const int x0 = blockIdx.x * AT_W, ry = blockIdx.y % step, tyd = blockIdx.y / step;
    {   // threads 0..127 stage footprint row 2k, threads 128..255 row 2k+1 (span <= 128 for step <= 16)
        const int ex = threadIdx.x & 127;
        const int px = x0 - 2 * step + ex;
        for (int k = 0; k < (AT_ROWS + 4) / 2; ++k) {
            const int ey = 2 * k + (threadIdx.x >> 7);
            const int yd = tyd * AT_ROWS - 2 + ey; // row index within the class
            const int py = yd * step + ry;
            if (ex < span) {
                const int e = ey * span + ex;
                if (px >= 0 && px < W && yd >= 0 && py < H) {
                    const int pi = py * W + px;
                    const float4 gv = g4[pi];
                    const float4 cv = in_c4[pi];
                    s_c[e] = cv;
                    s_l[e] = luminance(xyz(cv));
                    s_g[e] = gv;
                    // (a sky pixel is skipped by every tap that reaches it, whatever its other tests say -- the reference's
                    // `continue`s have no side effects --, so it is marked like a pixel outside the image, once, here,
                    // instead of being re-derived by each of the up to 25 taps that read it: 90 -> 86 us per pass)
                    s_o[e] = is_sky(gv.w, xyz(gv), sky) ? AT_OUTSIDE : (uo ? object_id[pi] : 0);
                } else {
                    s_o[e] = AT_OUTSIDE;
                }
            }
        }
    }
    __syncthreads();
    const int ti = threadIdx.x & (AT_W - 1), tj = threadIdx.x >> 6;
    const int x = x0 + ti, y = (tyd * AT_ROWS + tj) * step + ry;
    if (x >= W || y >= H)
        return;
    const int idx = y * W + x;
    const int ce = (tj + 2) * span + 2 * step + ti;
    const float4 c4 = s_c[ce];
    const float4 g = s_g[ce];
    const f3 cc = xyz(c4), cn = xyz(g);
    const float cd = g.w, cvar = c4.w;
    const int cobj = uo ? s_o[ce] : -1;
    const float clum = s_l[ce];
    f3 res = cc;
    float res_var = cvar;
    if (!is_sky(cd, cn, sky)) {
        const float var_scale = sqrt_ieee(max_(cvar, 1e-6f));
        const float asl = sigma_lum * (1.0f + var_scale * 2.0f);
        const float inv_sl2 = 1.0f / (2.0f * asl * asl + 1e-6f);
        const float fast_k = -inv_sl2 * 1.44269504f; // (FAST: exp(-x) = exp2(x * -log2 e), the constant folded once per pixel)
        f3 sum = mk3(0.0f);
        float sum_var = 0.0f, total_w = 0.0f;
        // The reference's chain of early `continue`s (denoiser.cuh:686-721) as ONE predicate per tap: every operand of a tap is
        // loaded unconditionally (an entry outside the image holds whatever the LDS held: its values take part in comparisons
        // whose results the first conjunct discards), so the LDS reads of a tap do not wait for its branches.
        // Relative-depth test `dd / max_d > edt`: q~ = dd * v_rcp_f32(max_d) is within 2^-22 of the quotient, the correctly
        // rounded quotient within 2^-24; outside edt (1 +- 2^-20) both sides of the comparison agree, and the division itself
        // runs only for a wave in which some tap lands in that band (or compares NaNs, or has a depth beyond 2^120, whose
        // reciprocal is not v_rcp_f32's business).
        // (Measured and dropped: the tap's contribution as selects too -- straight-line code over the 25 taps, all reads in
        // flight -- 125-131 us per pass against 87: the exponential of every rejected tap costs more than the branch saves.)
        const float e_lo = min_(edt * (1.0f - 0x1p-20f), edt * (1.0f + 0x1p-20f)), e_hi = max_(edt * (1.0f - 0x1p-20f), edt * (1.0f + 0x1p-20f));
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int ne = ce + dy * span + dx * step; // (x + dx*step, y + dy*step)
                const int nobj = s_o[ne];
                const float4 ng = s_g[ne];
                const float4 nc4 = s_c[ne];
                const float nlum = s_l[ne];
                bool ok = nobj != AT_OUTSIDE; // nx < 0 || nx >= W || ny < 0 || ny >= H
                ok = ok && !(uo && cobj != nobj && cobj >= 0 && nobj >= 0);
                // (hardware maximum: equal to max_ -- CUDA's fmaxf -- for every pair of operands that can reach the division
                // below: a NaN operand is ignored by both, and the one case in which they differ, the sign of max(+0, -0),
                // fails `max_d > 1e-6f` either way)
                const float max_d = __builtin_fmaxf(cd, ng.w);
                const float dd = __builtin_fabsf(cd - ng.w);
                const float qa = dd * __builtin_amdgcn_rcpf(max_d);
                bool far = qa > e_hi;
                const bool unsure = ok && max_d > 1e-6f && ((!(qa > e_hi) && !(qa < e_lo)) || !(max_d < 0x1p120f));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(unsure) != 0ull, 0))
                    far = unsure ? (dd / max_d > edt) : far;
                ok = ok && !(max_d > 1e-6f && far);
                const f3 nn = xyz(ng);
                ok = ok && !(dot(cn, nn) < ent); // (a sky neighbour carries AT_OUTSIDE: staging)
                if (ok) {
                    const f3 nc = xyz(nc4);
                    const float ld = __builtin_fabsf(clum - nlum);
                    const float wl = FAST ? __builtin_amdgcn_exp2f(ld * ld * fast_k) : det_exp(-ld * ld * inv_sl2);
                    // atrous_kernel[k] = (a*b)/256 with a,b in {1,4,6}: products and the /256 are exact in fp32
                    const float weight = (KW[dy + 2] * KW[dx + 2] / 256.0f) * wl;
                    sum = sum + nc * weight;
                    sum_var += nc4.w * weight;
                    total_w += weight;
                }
            }
        if (!(total_w < 1e-6f)) {
            const float inv_w = 1.0f / total_w;
            res = sum * inv_w;
            res_var = sum_var * inv_w;
        }
    }
    if (LAST) {
        st3(out3, idx, res);
        if (rgb8) { // NULL when bloom / up-scale follow and tonemap their own result
            unsigned char r, gg, b;
            tonemap_pixel(res, r, gg, b);
            const size_t o = ((size_t)(H - 1 - y) * W + x) * 3;
            rgb8[o] = r;
            rgb8[o + 1] = gg;
            rgb8[o + 2] = b;
        }
    } else {
        out_c4[idx] = mk4(res, res_var);
    }
}

// ------------------------------------------------------------------ stand-alone passes for atrous_iterations == 0
__global__ __launch_bounds__(256) void c4_to_output_kernel(const float4 *__restrict__ in_c4, int W, int H,
                                                           float *__restrict__ out3, unsigned char *__restrict__ rgb8) {
    PT_PIXEL_XY
    const f3 res = xyz(in_c4[idx]);
    st3(out3, idx, res);
    if (!rgb8)
        return;
    unsigned char r, g, b;
    tonemap_pixel(res, r, g, b);
    const size_t o = ((size_t)(H - 1 - y) * W + x) * 3;
    rgb8[o] = r;
    rgb8[o + 1] = g;
    rgb8[o + 2] = b;
}

} // namespace pt
