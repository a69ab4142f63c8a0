/*
 * oracle/denoiser_oracle.cpp -- TEST INFRASTRUCTURE (CPU oracle, never linked into the product).
 * PARITY UNPINNED (see ptrt_oracle.cpp): the reference ships no vectors for this stage either.
 *
 * Restates the post-process that follows the path tracer when the denoiser is enabled
 * (SURVEY 8(f) rank 1):
 *   motion_vector_kernel          src/pathtracer/rendering/denoiser_kernels.cuh:33-68
 *   firefly_suppression_kernel    src/pathtracer/rendering/denoiser.cuh:376-424
 *   init_moments_kernel           denoiser.cuh:751-763
 *   temporal_accumulation_kernel  denoiser.cuh:426-584  (+ edge-aware bilinear taps 231-374)
 *   estimate_variance_kernel      denoiser.cuh:586-648
 *   atrous_filter_kernel          denoiser.cuh:650-749
 *   Denoiser::denoiseChannel / denoise (non-split path)  denoiser.cuh:884-1064
 * wired as in Scene::render_to_device (scene/scene.cuh:1100-1127).
 *
 * One deliberate deviation, stated because the reference is not deterministic there:
 * denoiseChannel launches temporal_accumulation_kernel with out_mean == current_color == d_ping
 * (denoiser.cuh:908-910), so threads read 3x3 neighbourhoods of a buffer other threads are
 * overwriting.  This restatement (and the HIP kernels) read the PRE-kernel image everywhere
 * ("snapshot" semantics), which is the result the race gives when every read precedes every write.
 *
 * Same arithmetic contract as the path: -ffp-contract=off, dot() fused as in ptrt_oracle.cpp,
 * __expf -> dm_exp (detmath.h), sqrtf and / IEEE.
 */
#include "detmath.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float a, float b, float c) { return V3{a, b, c}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
inline float dot(V3 a, V3 b) { return dm_fma(a.z, b.z, dm_fma(a.y, b.y, a.x * b.x)); }
inline V3 max3(V3 a, V3 b) { return v3(dm_max(a.x, b.x), dm_max(a.y, b.y), dm_max(a.z, b.z)); }
inline V3 min3(V3 a, V3 b) { return v3(dm_min(a.x, b.x), dm_min(a.y, b.y), dm_min(a.z, b.z)); }
inline float luminance(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
inline int clamp_int(int v, int a, int b) { return v < a ? a : (v > b ? b : v); }
inline float clampf(float x, float a, float b) { return dm_min(dm_max(x, a), b); }
inline bool is_sky(float depth, V3 n, float thr) { return (depth > thr) || (dot(n, n) < 0.1f); } // denoiser.cuh:119-122

struct Img { // views into the caller's arrays
    const V3 *c;
};

// denoiser.cuh:209-229
inline bool is_edge_discontinuity(float d0, float d1, V3 n0, V3 n1, int obj0, int obj1, float depth_threshold,
                                  float normal_threshold, bool use_obj_id) {
    if (use_obj_id && obj0 != obj1 && obj0 >= 0 && obj1 >= 0)
        return true;
    float max_d = dm_max(d0, d1);
    float depth_diff = fabsf(d0 - d1);
    if (max_d > 1e-6f && depth_diff / max_d > depth_threshold)
        return true;
    float n_dot = dot(n0, n1);
    if (n_dot < normal_threshold)
        return true;
    return false;
}

struct Taps { // the four bilinear taps shared by the edge-aware samplers (denoiser.cuh:239-292)
    int idx[4];
    float w[4];
    bool valid[4];
    float total_w;
    int nearest;
};
inline Taps make_taps(const float *depth_buf, const V3 *normal_buf, const int *obj_buf, int width, int height, float u,
                      float v, float center_depth, V3 center_normal, int center_obj, float edt, float ent, bool use_obj_id) {
    bool use_obj = use_obj_id && (obj_buf != nullptr);
    float fx = u - 0.5f, fy = v - 0.5f;
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    int x1 = x0 + 1, y1 = y0 + 1;
    float sx = fx - x0, sy = fy - y0;
    x0 = clamp_int(x0, 0, width - 1);
    y0 = clamp_int(y0, 0, height - 1);
    x1 = clamp_int(x1, 0, width - 1);
    y1 = clamp_int(y1, 0, height - 1);
    Taps t;
    t.idx[0] = y0 * width + x0;
    t.idx[1] = y0 * width + x1;
    t.idx[2] = y1 * width + x0;
    t.idx[3] = y1 * width + x1;
    const float bw[4] = {(1.0f - sx) * (1.0f - sy), sx * (1.0f - sy), (1.0f - sx) * sy, sx * sy};
    for (int k = 0; k < 4; ++k) {
        const int o = use_obj ? obj_buf[t.idx[k]] : -1;
        t.valid[k] = !is_edge_discontinuity(center_depth, depth_buf[t.idx[k]], center_normal, normal_buf[t.idx[k]],
                                            center_obj, o, edt, ent, use_obj);
        t.w[k] = t.valid[k] ? bw[k] : 0.0f;
    }
    t.total_w = t.w[0] + t.w[1] + t.w[2] + t.w[3];
    t.nearest = clamp_int((int)floorf(v), 0, height - 1) * width + clamp_int((int)floorf(u), 0, width - 1);
    return t;
}
inline V3 sample_v3(const V3 *buf, const Taps &t) {
    if (t.total_w < 1e-6f) {
        for (int k = 0; k < 4; ++k)
            if (t.valid[k])
                return buf[t.idx[k]];
        return buf[t.nearest];
    }
    return (buf[t.idx[0]] * t.w[0] + buf[t.idx[1]] * t.w[1] + buf[t.idx[2]] * t.w[2] + buf[t.idx[3]] * t.w[3]) *
           (1.0f / t.total_w);
}
inline float sample_f(const float *buf, const Taps &t) {
    if (t.total_w < 1e-6f) {
        for (int k = 0; k < 4; ++k)
            if (t.valid[k])
                return buf[t.idx[k]];
        return buf[t.nearest];
    }
    return (buf[t.idx[0]] * t.w[0] + buf[t.idx[1]] * t.w[1] + buf[t.idx[2]] * t.w[2] + buf[t.idx[3]] * t.w[3]) *
           (1.0f / t.total_w);
}

const float ATROUS[25] = {1.0f / 256.0f,  4.0f / 256.0f,  6.0f / 256.0f,  4.0f / 256.0f,  1.0f / 256.0f,
                          4.0f / 256.0f,  16.0f / 256.0f, 24.0f / 256.0f, 16.0f / 256.0f, 4.0f / 256.0f,
                          6.0f / 256.0f,  24.0f / 256.0f, 36.0f / 256.0f, 24.0f / 256.0f, 6.0f / 256.0f,
                          4.0f / 256.0f,  16.0f / 256.0f, 24.0f / 256.0f, 16.0f / 256.0f, 4.0f / 256.0f,
                          1.0f / 256.0f,  4.0f / 256.0f,  6.0f / 256.0f,  4.0f / 256.0f,  1.0f / 256.0f};

} // namespace

extern "C" {

/* DenoiserSettings (denoiser.cuh:36-73), the fields the non-split path reads */
struct oracle_denoiser_settings {
    float tau, min_alpha, max_history, sigma_luminance, sigma_normal, sigma_depth;
    int32_t atrous_iterations;
    float clamp_scale, firefly_threshold;
    float depth_reject_absolute, depth_reject_relative, normal_reject_threshold, sky_depth_threshold;
    float edge_depth_threshold, edge_normal_threshold;
    int32_t use_object_ids, enable_firefly_suppression;
};

/* persistent state of `class Denoiser` (denoiser.cuh:781-806), owned by the caller */
struct oracle_denoiser_state {
    int32_t width, height, first_frame;
    float *history_mean, *history_m2, *history_length; /* 3,3,1 floats per pixel */
    float *history_normal, *history_depth;
    int32_t *history_object_id;
};

/* Camera::random_in_unit_disk_hash (camera.cuh:55-72); cosf/sinf -> dm_cos/dm_sin (detmath.h) */
static V3 random_in_unit_disk_hash(uint32_t x, uint32_t y) {
    uint32_t seed = (x * 1973u) ^ (y * 9277u) ^ 0x9e3779b9u;
    seed ^= seed >> 17;
    seed *= 0xed5ad4bbu;
    seed ^= seed >> 11;
    seed *= 0xac4c1b51u;
    seed ^= seed >> 15;
    seed *= 0x31848babu;
    seed ^= seed >> 14;
    float r1 = ((seed & 0xFFFFu) + 0.5f) / 65536.0f;
    float r2 = (((seed * 0x343fdu + 0xc0f5u) & 0xFFFFu) + 0.5f) / 65536.0f;
    float r = sqrtf(r1);
    float phi = 6.2831853f * r2;
    return v3(r * dm_cos(phi), r * dm_sin(phi), 0.0f);
}

/* motion_vector_kernel.  cam19 = origin, lower_left_corner, horizontal, vertical, u, v (18 floats) + lens_radius;
 * prev_view_proj = 16 floats, column-major as mat4 stores them. */
void oracle_motion_vectors(const float *depth, int W, int H, const float *cam19, const float *pvp, float *out_mv2) {
    const float *cam12 = cam19;
    const V3 origin = v3(cam12[0], cam12[1], cam12[2]), llc = v3(cam12[3], cam12[4], cam12[5]),
             hor = v3(cam12[6], cam12[7], cam12[8]), ver = v3(cam12[9], cam12[10], cam12[11]);
    const V3 cu = v3(cam19[12], cam19[13], cam19[14]), cv = v3(cam19[15], cam19[16], cam19[17]);
    const float lens_radius = cam19[18];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t idx = (size_t)y * W + x;
            float d = depth[idx];
            if (d >= 1e29f) { // DENOISER_SKY_DEPTH_THRESHOLD
                out_mv2[idx * 2] = out_mv2[idx * 2 + 1] = 0.0f;
                continue;
            }
            float u = (x + 0.5f) / W;
            float v = (y + 0.5f) / H;
            // cam.get_ray(u, 1-v) (camera.cuh:173-205): lens_radius <= 0 -> get_ray_simple, else the device
            // branch: a lens sample hashed from (s, t)
            const float s = u, t = 1.0f - v;
            V3 ray_o = origin;
            V3 rd = llc + hor * s + ver * t - origin;
            if (lens_radius > 0) {
                uint32_t hx = (uint32_t)(s * 10000.0f) + (uint32_t)(t * 5000.0f);
                uint32_t hy = (uint32_t)(t * 10000.0f) + (uint32_t)(s * 5000.0f);
                V3 disk = random_in_unit_disk_hash(hx, hy) * lens_radius;
                V3 offset = cu * disk.x + cv * disk.y;
                rd = llc + hor * s + ver * t - origin - offset;
                ray_o = origin + offset;
            }
            float len = sqrtf(dot(rd, rd));
            V3 dir = (len > 0) ? v3(rd.x / len, rd.y / len, rd.z / len) : v3(0.0f);
            V3 wp = ray_o + dir * d;
            // mat4 * vec4, column-major (mat4.cuh:269-274)
            float cx = pvp[0] * wp.x + pvp[4] * wp.y + pvp[8] * wp.z + pvp[12] * 1.0f;
            float cy = pvp[1] * wp.x + pvp[5] * wp.y + pvp[9] * wp.z + pvp[13] * 1.0f;
            float cw = pvp[3] * wp.x + pvp[7] * wp.y + pvp[11] * wp.z + pvp[15] * 1.0f;
            float ndc_x = cx / cw, ndc_y = cy / cw;
            float prev_u = (ndc_x + 1.0f) * 0.5f;
            float prev_v = (1.0f - ndc_y) * 0.5f;
            out_mv2[idx * 2] = u - prev_u;
            out_mv2[idx * 2 + 1] = v - prev_v;
        }
}

/* One Denoiser::denoise call, non-split path.  color/normal: 3 floats per pixel; motion: 2. */
void oracle_denoise(const oracle_denoiser_settings *S, oracle_denoiser_state *st, const float *color_in,
                    const float *normal_in, const float *depth, const float *motion, const int32_t *object_id,
                    float *out_color) {
    const int W = st->width, H = st->height;
    const size_t n = (size_t)W * H;
    const V3 *src = reinterpret_cast<const V3 *>(color_in);
    const V3 *normal = reinterpret_cast<const V3 *>(normal_in);
    V3 *hmean = reinterpret_cast<V3 *>(st->history_mean), *hm2 = reinterpret_cast<V3 *>(st->history_m2);
    float *hlen = st->history_length;
    V3 *hnormal = reinterpret_cast<V3 *>(st->history_normal);
    const bool use_obj_id = S->use_object_ids && object_id != nullptr;
    const float sky = S->sky_depth_threshold;

    if (st->first_frame) { // denoiser.cuh:997-1009
        memcpy(hnormal, normal, n * sizeof(V3));
        memcpy(st->history_depth, depth, n * sizeof(float));
        if (object_id)
            memcpy(st->history_object_id, object_id, n * sizeof(int32_t));
    }

    // ---- firefly suppression (denoiser.cuh:376-424) or plain copy
    std::vector<V3> ping(n), pong(n), new_mean(n);
    std::vector<float> var_a(n), var_b(n), new_len(n);
    if (S->enable_firefly_suppression) {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int idx = y * W + x;
                V3 center = src[idx];
                if (is_sky(depth[idx], normal[idx], sky)) {
                    ping[idx] = center;
                    continue;
                }
                V3 mx = v3(0.0f);
                bool any = false;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (dx == 0 && dy == 0)
                            continue;
                        int nx = x + dx, ny = y + dy;
                        if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
                            mx = max3(mx, src[ny * W + nx]);
                            any = true;
                        }
                    }
                if (any) {
                    V3 c = min3(center, mx * 1.25f);
                    c = min3(c, v3(10.0f));
                    ping[idx] = c;
                } else {
                    ping[idx] = center;
                }
            }
    } else {
        memcpy(ping.data(), src, n * sizeof(V3));
    }

    if (st->first_frame) { // denoiser.cuh:899-905
        memcpy(hmean, ping.data(), n * sizeof(V3));
        for (size_t i = 0; i < n; ++i) {
            hm2[i] = ping[i] * ping[i];
            hlen[i] = 1.0f;
        }
    }

    // ---- temporal accumulation (denoiser.cuh:426-584), snapshot semantics
    const float *pdepth = st->history_depth;
    const int32_t *pobj = st->history_object_id;
    const bool use_obj = use_obj_id && (object_id != nullptr) && (pobj != nullptr);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int idx = y * W + x;
            V3 cur_c = ping[idx];
            float d = depth[idx];
            V3 nn0 = normal[idx];
            int obj_id = use_obj ? object_id[idx] : -1;
            if (is_sky(d, nn0, sky)) {
                new_mean[idx] = cur_c;
                pong[idx] = cur_c * cur_c;
                new_len[idx] = 1.0f;
                continue;
            }
            V3 nmean = v3(0.0f), nm2 = v3(0.0f);
            int ncount = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int nx = clamp_int(x + dx, 0, W - 1), ny = clamp_int(y + dy, 0, H - 1);
                    int ni = ny * W + nx;
                    V3 nc = ping[ni];
                    int no = use_obj ? object_id[ni] : -1;
                    bool same = !is_edge_discontinuity(d, depth[ni], nn0, normal[ni], obj_id, no,
                                                       S->edge_depth_threshold, S->edge_normal_threshold, use_obj);
                    if (same) {
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
            float inv_n = 1.0f / (float)ncount;
            nmean = nmean * inv_n;
            nm2 = nm2 * inv_n;
            V3 nvar = max3(nm2 - nmean * nmean, v3(0.0f));
            V3 nstd = v3(sqrtf(nvar.x), sqrtf(nvar.y), sqrtf(nvar.z));
            V3 soft_min = nmean - nstd * S->clamp_scale;
            V3 soft_max = nmean + nstd * S->clamp_scale;

            float mvx = motion[idx * 2], mvy = motion[idx * 2 + 1];
            float prev_u = (float)x + 0.5f - mvx * W;
            float prev_v = (float)y + 0.5f - mvy * H;
            bool valid = true;
            if (prev_u < 0.5f || prev_v < 0.5f || prev_u >= (float)(W - 0.5f) || prev_v >= (float)(H - 0.5f))
                valid = false;
            V3 hist_mean = v3(0.0f), hist_m2 = v3(0.0f);
            float hist_len = 0.0f;
            if (valid) {
                Taps t = make_taps(pdepth, hnormal, pobj, W, H, prev_u, prev_v, d, nn0, obj_id, S->edge_depth_threshold,
                                   S->edge_normal_threshold, use_obj);
                hist_mean = sample_v3(hmean, t);
                hist_m2 = sample_v3(hm2, t);
                hist_len = sample_f(hlen, t);
                float hist_d = sample_f(pdepth, t);
                if (use_obj && pobj != nullptr) {
                    int hist_obj = pobj[t.nearest];
                    if (hist_obj != obj_id)
                        valid = false;
                }
                float dad = fabsf(d - hist_d);
                if (dad > S->depth_reject_absolute || dad > S->depth_reject_relative * dm_max(1e-6f, d))
                    valid = false;
                V3 hist_n = hnormal[t.nearest];
                if (dot(nn0, hist_n) < S->normal_reject_threshold)
                    valid = false;
            }
            if (valid)
                hist_mean = min3(max3(hist_mean, soft_min), soft_max);
            float alpha = 1.0f, nlen = 1.0f;
            if (valid) {
                V3 var = max3(hist_m2 - (hist_mean * hist_mean), v3(0.0f));
                float std_approx = (sqrtf(var.x) + sqrtf(var.y) + sqrtf(var.z)) * (1.0f / 3.0f);
                float variance_alpha = std_approx / (std_approx + S->tau);
                float history_alpha = 1.0f / (hist_len + 1.0f);
                alpha = clampf(dm_max(variance_alpha, history_alpha), S->min_alpha, 1.0f);
                nlen = dm_min(hist_len + 1.0f, S->max_history);
            }
            new_mean[idx] = hist_mean * (1.0f - alpha) + cur_c * alpha;
            pong[idx] = hist_m2 * (1.0f - alpha) + (cur_c * cur_c) * alpha;
            new_len[idx] = nlen;
        }
    // denoiser.cuh:922-930: results become the history
    memcpy(hmean, new_mean.data(), n * sizeof(V3));
    memcpy(hm2, pong.data(), n * sizeof(V3));
    memcpy(hlen, new_len.data(), n * sizeof(float));
    ping = new_mean;

    // ---- variance estimate (denoiser.cuh:586-648); note: use_object_ids, not use_obj_id (932-936)
    {
        const bool uo = S->use_object_ids && (object_id != nullptr);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int idx = y * W + x;
                float d = depth[idx];
                V3 nn0 = normal[idx];
                int obj = uo ? object_id[idx] : -1;
                if (is_sky(d, nn0, sky)) {
                    var_a[idx] = 0.0f;
                    continue;
                }
                V3 c = ping[idx], cm2 = hm2[idx];
                float hl = hlen[idx];
                V3 var = max3(cm2 - (c * c), v3(0.0f));
                float reliability = dm_min(hl * 0.25f, 1.0f);
                float boost = 1.0f + (1.0f - reliability) * 3.0f;
                V3 smean = v3(0.0f), sm2 = v3(0.0f);
                int count = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int nx = clamp_int(x + dx, 0, W - 1), ny = clamp_int(y + dy, 0, H - 1);
                        int ni = ny * W + nx;
                        if (uo && object_id[ni] != obj)
                            continue;
                        V3 nc = ping[ni];
                        smean = smean + nc;
                        sm2 = sm2 + nc * nc;
                        count++;
                    }
                const float inv = 1.0f / (float)count;
                smean = smean * inv;
                sm2 = sm2 * inv;
                V3 svar = max3(sm2 - smean * smean, v3(0.0f));
                V3 cv = max3(var * boost, svar);
                var_a[idx] = 0.2126f * cv.x + 0.7152f * cv.y + 0.0722f * cv.z;
            }
    }

    // ---- a-trous wavelet iterations (denoiser.cuh:650-749, 938-961)
    {
        const bool uo = S->use_object_ids && (object_id != nullptr);
        const int steps[5] = {1, 2, 4, 8, 16};
        const int iters = S->atrous_iterations < 5 ? S->atrous_iterations : 5;
        std::vector<V3> *in = &ping, *out = &pong;
        std::vector<float> *vin = &var_a, *vout = &var_b;
        for (int it = 0; it < iters; ++it) {
            const int step = steps[it];
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int idx = y * W + x;
                    V3 cc = (*in)[idx], cn = normal[idx];
                    float cd = depth[idx];
                    int cobj = uo ? object_id[idx] : -1;
                    float cvar = (*vin)[idx];
                    float clum = luminance(cc);
                    if (is_sky(cd, cn, sky)) {
                        (*out)[idx] = cc;
                        (*vout)[idx] = cvar;
                        continue;
                    }
                    float var_scale = sqrtf(dm_max(cvar, 1e-6f));
                    float asl = S->sigma_luminance * (1.0f + var_scale * 2.0f);
                    float inv_sl2 = 1.0f / (2.0f * asl * asl + 1e-6f);
                    V3 sum = v3(0.0f);
                    float sum_var = 0.0f, total_w = 0.0f;
                    for (int dy = -2; dy <= 2; ++dy)
                        for (int dx = -2; dx <= 2; ++dx) {
                            int k = (dy + 2) * 5 + (dx + 2);
                            int nx = x + dx * step, ny = y + dy * step;
                            if (nx < 0 || nx >= W || ny < 0 || ny >= H)
                                continue;
                            int ni = ny * W + nx;
                            if (uo) {
                                int nobj = object_id[ni];
                                if (cobj != nobj && cobj >= 0 && nobj >= 0)
                                    continue;
                            }
                            float nd = depth[ni];
                            float max_d = dm_max(cd, nd);
                            float dd = fabsf(cd - nd);
                            if (max_d > 1e-6f && dd / max_d > S->edge_depth_threshold)
                                continue;
                            V3 nnn = normal[ni];
                            if (dot(cn, nnn) < S->edge_normal_threshold)
                                continue;
                            if (is_sky(nd, nnn, sky))
                                continue;
                            V3 nc = (*in)[ni];
                            float nvar = (*vin)[ni];
                            float ld = fabsf(clum - luminance(nc));
                            float wl = dm_exp(-ld * ld * inv_sl2);
                            float weight = ATROUS[k] * wl;
                            sum = sum + nc * weight;
                            sum_var += nvar * weight;
                            total_w += weight;
                        }
                    if (total_w < 1e-6f) {
                        (*out)[idx] = cc;
                        (*vout)[idx] = cvar;
                    } else {
                        float inv_w = 1.0f / total_w;
                        (*out)[idx] = sum * inv_w;
                        (*vout)[idx] = sum_var * inv_w;
                    }
                }
            std::swap(in, out);
            std::swap(vin, vout);
        }
        memcpy(out_color, in->data(), n * sizeof(V3));
    }

    // denoiser.cuh:1049-1061
    memcpy(hnormal, normal, n * sizeof(V3));
    memcpy(st->history_depth, depth, n * sizeof(float));
    if (object_id)
        memcpy(st->history_object_id, object_id, n * sizeof(int32_t));
    st->first_frame = 0;
}

} // extern "C"
