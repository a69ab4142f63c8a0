// pt_wavefront.hip.h -- the render loop as wavefront stages (BASELINE configs[2]/[3]: scenes whose
// BLASes are real trees).
//
// Why: in the megakernel a wave owns 64 pixels, so a traversal phase lasts as long as the LONGEST of
// its <= 64 rays while most of them end after a few nodes (showcase scene, measured with
// PT_TRAV_STATS: 39 node iterations per phase for 8.3 nodes per ray -> 17 % of the lanes busy in the
// closest-hit loops, 6 % in the shadow loops, which see ~10 rays per wave).  No in-wave scheduling
// can fix that: the wave has no other rays to give to its idle lanes.  Here the rays of the WHOLE
// frame are one pool:
//
//   shade  (one thread per path)   resolve the previous light sample with its visibility, finish /
//                                  regenerate paths, shade the hit of the last extension ray (emission,
//                                  light sample, BSDF sample, roulette), write the next extension ray
//                                  and/or shadow ray of the path into planar HBM buffers
//   trace  (persistent waves)      every wave scans its stripe of the path-state words, compacts the
//                                  paths that have a ray (ballot + prefix sum) into a small LDS ring, and
//                                  whenever fetch_min lanes are idle those lanes take the next rays from
//                                  the ring -- shadow (any-hit) and extension (closest-hit) rays mixed in
//                                  one launch, one code path for both.  A lane walks its ray through the
//                                  meshes whose root box it hits, in TLAS order, exactly as one thread of
//                                  the reference does (bvh_trace_tlas / bvh_any_hit_tlas,
//                                  intersection.cuh:438-605), so no merge step and no tie rule is needed.
//
// and the frame is  shade(first) ; { trace ; shade } x spp*(max_depth+1).  No queue counters and no
// atomics: the "queue" is the state word array read in stripe order; launches past the last live path
// return at once (live flag per iteration).
//
// Bits are those of the megakernel by construction: per pixel, the random numbers are drawn in the
// same order (light sample, BSDF sample, roulette, next sample's lens sample), every float is
// combined in the same order (the light sample's contribution is formed at the hit -- BSDF, MIS
// weight, throughput -- and ADDED when its visibility is known, before anything else touches the
// path's radiance), and the traversal of one ray is the per-thread algorithm.
// tests/test_wavefront_gpu.py compares every buffer with the oracle and with the megakernel.
#pragma once
#include "pt_render.hip.h"

namespace pt {

struct WfParams {
    uint32_t *st;   // per path: sample (bits 0-7), bounce (8-15), WF_* flags
    float *ray;     // 6 planes: origin xyz, direction xyz of the extension ray
    float4 *hit;    // {t, t_local, mesh, slot} of the extension ray
    float *thr;     // 3 planes each: throughput, radiance of the current sample, sum over finished samples,
    float *acc;     //                contribution of the pending light sample
    float *avg;
    float *pend;
    float *sh;      // 7 planes: shadow ray origin xyz, direction xyz, tmax
    uint32_t *occ;  // shadow ray result
    uint32_t *live; // [iteration] != 0: some path is still alive after that iteration's shade
    int n_items;    // tiles * 64 (tile-ordered: path q = tile*64 + lane, lane = 8*row + column of the 8x8 tile)
    int iter;       // index of this shade / trace (0 = the regenerate-only first shade)
    int fetch_min;
};
constexpr uint32_t WF_SPEC = 1u << 16, WF_PREV_SPEC = 1u << 17, WF_EXT = 1u << 18, WF_SHADOW = 1u << 19,
                   WF_ENDED = 1u << 20, WF_DONE = 1u << 21, WF_PVALID = 1u << 22;
constexpr int WF_RING = 256; // ring entries per wave (u32): < 64 waiting + at most 128 from one chunk

// ---------------------------------------------------------------------------------------------
// trace: persistent waves, per-lane refill.  Preconditions (checked by the host): single-leaf TLAS
// with at most 64 meshes, stack_entries >= deepest BLAS.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void wf_trace_kernel(const KParams K,
                                                                                                  const WfParams W) {
    if (W.iter > 0 && W.live[W.iter - 1] == 0u)
        return;
    extern __shared__ uint2 lds_raw[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int per_wave = K.stack_entries * 64 + WF_RING / 2; // in uint2
    uint2 *wbase = lds_raw + (size_t)wib * per_wave;
    LdsStack stk{wbase + lane};
    uint32_t *ring = (uint32_t *)(wbase + K.stack_entries * 64);
    const int NW = gridDim.x * 4;
    const int n_chunks = (W.n_items + 63) >> 6;
    int chunk = blockIdx.x * 4 + wib;
    int ring_head = 0, ring_n = 0; // wave-uniform
    const int2 tl = K.tlas_leaves[~K.tlas_root_ref];
    const int M = tl.y;

    bool busy = false, active = false, anyq = false, xf = false, found = false;
    int q = 0, cur = 0, sp = 0, sb = -1, mi = 0, gmesh = -1, gslot = -1;
    unsigned long long mask = 0ull;
    float ds = 1.0f, tb = T_FAR, tm = T_FAR, gt = T_FAR, gtl = T_FAR;
    RayO pr = make_ray(mk3(0.0f), mk3(0.0f, 0.0f, 1.0f));

    auto pop = [&]() {
        active = false;
        while (sp > 0) {
            --sp;
            int ref;
            float tE;
            stk.pop(sp, ref, tE);
            if (tE < tb) {
                cur = ref;
                active = true;
                break;
            }
        }
    };
    auto world_ray = [&](f3 &o, f3 &d) {
        const float *src = anyq ? W.sh : W.ray;
        const size_t n = (size_t)W.n_items;
        o = mk3(src[q], src[n + q], src[2 * n + q]);
        d = mk3(src[3 * n + q], src[4 * n + q], src[5 * n + q]);
    };

    for (;;) {
        // ---- refill: idle lanes take the next rays of this wave's stripe
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const int n_idle = __builtin_popcountll(idle);
        if (n_idle >= W.fetch_min || n_idle == 64) {
            while (ring_n < n_idle && chunk < n_chunks) {
                const int i = chunk * 64 + lane;
                const uint32_t f = (i < W.n_items) ? W.st[i] : 0u;
                const bool shq = (f & WF_SHADOW) != 0u, exq = (f & WF_EXT) != 0u;
                const unsigned long long b1 = __builtin_amdgcn_ballot_w64(shq);
                if (shq)
                    ring[(ring_head + ring_n + lane_prefix(b1)) & (WF_RING - 1)] = (uint32_t)i * 2u + 1u;
                ring_n += __builtin_popcountll(b1);
                const unsigned long long b2 = __builtin_amdgcn_ballot_w64(exq);
                if (exq)
                    ring[(ring_head + ring_n + lane_prefix(b2)) & (WF_RING - 1)] = (uint32_t)i * 2u;
                ring_n += __builtin_popcountll(b2);
                chunk += NW;
            }
            wave_lds_order();
            const int take = n_idle < ring_n ? n_idle : ring_n;
            if (take > 0) {
                const int rank = lane_prefix(idle);
                if (!busy && rank < take) {
                    const uint32_t e = ring[(ring_head + rank) & (WF_RING - 1)];
                    q = (int)(e >> 1);
                    anyq = (e & 1u) != 0u;
                    f3 o, d;
                    world_ray(o, d);
                    tm = anyq ? W.sh[(size_t)6 * W.n_items + q] : T_FAR;
                    // root boxes: the TLAS node, then every mesh of its leaf (bvh_trace_tlas / bvh_any_hit_tlas)
                    const RayO w = make_ray(o, d);
                    float tE;
                    const bool in = slab(tlas_bmin(K), tlas_bmax(K), w, tm, tE);
                    mask = 0ull;
                    for (int i = 0; i < M; ++i) {
                        const int m = __builtin_amdgcn_readfirstlane(K.tlas_mesh_ids[tl.x + i]);
                        const MeshHead mh = load_mesh_head(K, m);
                        if (anyq && (mh.flags & 2))
                            continue;
                        bool hb;
                        if (mh.flags & 1) {
                            float s;
                            const RayO lr = local_ray(K, m, w, s);
                            hb = in && slab(mh.bmin, mh.bmax, lr, anyq ? tm * s : T_FAR, tE);
                        } else {
                            hb = in && slab(mh.bmin, mh.bmax, w, tm, tE);
                        }
                        mask |= hb ? (1ull << i) : 0ull;
                    }
                    pr = w;
                    xf = false;
                    busy = true;
                    active = false;
                    found = false;
                    sb = -1;
                    sp = 0;
                    gt = gtl = T_FAR;
                    gmesh = gslot = -1;
                }
                ring_head = (ring_head + take) & (WF_RING - 1);
                ring_n -= take;
            }
            wave_lds_order();
        }
        if (!__builtin_amdgcn_ballot_w64(busy))
            break; // nothing in flight, ring empty, stripe exhausted

        // ---- a lane between meshes: keep the mesh's closest hit, then the next mesh or the result
        if (busy && !active) {
            if (!anyq && sb >= 0) { // strict <: the earlier mesh keeps a tie (intersection.cuh:561)
                const float tw = xf ? tb / ds : tb;
                if (tw < gt) {
                    gt = tw;
                    gtl = tb;
                    gmesh = mi;
                    gslot = sb;
                }
            }
            if (found || mask == 0ull) {
                if (anyq)
                    W.occ[q] = found ? 1u : 0u;
                else
                    W.hit[q] = make_float4(gt, gtl, __int_as_float(gmesh), __int_as_float(gslot));
                busy = false;
            } else {
                const int i = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                mi = K.tlas_mesh_ids[tl.x + i];
                const float4 r0 = K.mesh_recs[mi * MESH_REC_F4], r1 = K.mesh_recs[mi * MESH_REC_F4 + 1];
                const bool nxf = (__float_as_int(r1.w) & 1) != 0;
                if (nxf || xf) { // leaving or entering a mesh's local space: rebuild the ray from the world ray
                    f3 o, d;
                    world_ray(o, d);
                    const RayO w = make_ray(o, d);
                    ds = 1.0f;
                    pr = nxf ? local_ray(K, mi, w, ds) : w;
                }
                xf = nxf;
                cur = __float_as_int(r0.w);
                tb = anyq ? (xf ? tm * ds : tm) : T_FAR;
                sb = -1;
                sp = 0;
                active = true;
            }
        }

        // ---- inner nodes (child-pair nodes; near child first, far child stacked with its entry distance)
        while (active && cur >= 0) {
            const float4 n0 = K.nodes[cur * 4 + 0], n1 = K.nodes[cur * 4 + 1], n2 = K.nodes[cur * 4 + 2],
                         n3 = K.nodes[cur * 4 + 3];
            float tL, tR;
            const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), pr, tb, tL);
            const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), pr, tb, tR);
            const int Lr = __float_as_int(n3.x), Rr = __float_as_int(n3.y);
            if (hL || hR) {
                const bool nearL = hL && (!hR || tL <= tR);
                if (nearL ? hR : hL) {
                    stk.push(sp, nearL ? Rr : Lr, nearL ? tR : tL);
                    ++sp;
                }
                cur = nearL ? Lr : Rr;
            } else {
                pop();
            }
        }
        // ---- leaf
        if (active) {
            const int2 lf = K.leaves[~cur];
            const float4 *tp = K.tris + (size_t)lf.x * 3;
            float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
            if (lf.y > 0) {
                p0 = tp[0];
                p1 = tp[1];
                p2 = tp[2];
            }
            for (int i = 0; i < lf.y; ++i) {
                const int nx = (i + 1 < lf.y) ? (i + 1) : i;
                const float4 q0 = tp[nx * 3 + 0], q1 = tp[nx * 3 + 1], q2 = tp[nx * 3 + 2];
                float t, u, v;
                if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), pr, tb, t, u, v)) {
                    if (anyq) {
                        found = true;
                    } else {
                        tb = t;
                        sb = lf.x + i;
                    }
                }
                p0 = q0;
                p1 = q1;
                p2 = q2;
            }
            if (found)
                active = false;
            else
                pop();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// shade: one thread per path.  The body is phases [A], [C], [E] of path_trace_kernel (pt_render.hip.h),
// i.e. scene_kernels.cuh:147-193 and tracePath (path_logic.cuh:782-899), cut at the two traces.
struct WfCounts {
    uint32_t n_ext = 0, n_shadow = 0, n_paths = 0, n_zero = 0;
    bool alive_after = false;
};
template <bool FULL> PT_DEV void wf_shade_path(const KParams &K, const WfParams &W, const int q, WfCounts &cn) {
    const bool first = W.iter == 0;
    const size_t N = (size_t)W.n_items;
    uint32_t &n_ext = cn.n_ext, &n_shadow = cn.n_shadow, &n_paths = cn.n_paths, &n_zero = cn.n_zero;
    bool &alive_after = cn.alive_after;
    if (q < W.n_items) {
        const int tile = q >> 6, l = q & 63;
        const int tx = tile % K.tiles_x, ty = tile / K.tiles_x;
        const int x = tx * 8 + (l & 7);
        const int yl = ty * 8 + (l >> 3);
        const bool inside = (x < K.width) && (yl < K.rows);
        const int y = global_row(yl, K.y0, K.il_period, K.il_phase);
        const size_t npix = K.rng_plane;
        const size_t idx = (size_t)yl * K.width + x;
        uint32_t st = first ? 0u : W.st[q];
        if (first && !inside) {
            W.st[q] = WF_DONE;
        } else if (!(st & WF_DONE)) {
            int s = (int)(st & 0xffu), bounce = (int)((st >> 8) & 0xffu);
            bool ray_spec = (st & WF_SPEC) != 0u, prev_was_specular = (st & WF_PREV_SPEC) != 0u;
            Rng rng;
            rng.d = K.rng[idx];
            rng.v0 = K.rng[npix + idx];
            rng.v1 = K.rng[2 * npix + idx];
            rng.v2 = K.rng[3 * npix + idx];
            rng.v3 = K.rng[4 * npix + idx];
            rng.v4 = K.rng[5 * npix + idx];
            f3 throughput = mk3(1.0f), acc = mk3(0.0f);
            f3 ro = mk3(0.0f), rd = mk3(0.0f);
            bool regen = first;
            uint32_t flags = 0u;
            if (!first) {
                throughput = mk3(W.thr[q], W.thr[N + q], W.thr[2 * N + q]);
                acc = mk3(W.acc[q], W.acc[N + q], W.acc[2 * N + q]);
                // the light sample of the previous vertex, now that its visibility is known
                if ((st & WF_SHADOW) && (st & WF_PVALID) && W.occ[q] == 0u)
                    acc = acc + mk3(W.pend[q], W.pend[N + q], W.pend[2 * N + q]);
                if (st & WF_ENDED) {
                    regen = true;
                } else if (st & WF_EXT) {
                    ro = mk3(W.ray[q], W.ray[N + q], W.ray[2 * N + q]);
                    rd = mk3(W.ray[3 * N + q], W.ray[4 * N + q], W.ray[5 * N + q]);
                    const float4 hr = W.hit[q];
                    Hit h;
                    h.t = hr.x;
                    h.t_local = hr.y;
                    h.u = h.v = 0.0f;
                    h.mesh = __float_as_int(hr.z);
                    h.slot = __float_as_int(hr.w);
                    ++n_ext;
                    bool end_path = false, want_shadow = false;
                    if (h.mesh < 0) {
                        if (bounce == 0 && s == 0) {
                            K.normal[idx * 3 + 0] = 0.0f;
                            K.normal[idx * 3 + 1] = 0.0f;
                            K.normal[idx * 3 + 2] = 0.0f;
                            K.depth[idx] = 1e30f;
                            K.object_id[idx] = -1;
                        }
                        if (K.use_sky) {
                            if (K.env) {
                                const float phi = det_atan2(rd.z, rd.x);
                                const float theta = det_acos(max_(-1.0f, min_(1.0f, rd.y)));
                                const float u = (phi + PI_F) * (1.0f / TWO_PI_F);
                                const float v = theta * (1.0f / PI_F);
                                acc = acc + throughput * tex2d_env(K.env, K.env_w, K.env_h, u, v);
                            } else {
                                const float t = 0.5f * (rd.y + 1.0f);
                                acc = acc + throughput * lerp(K.sky_bottom, K.sky_top, t);
                            }
                        } else {
                            acc = acc + throughput * mk3(0.0f);
                        }
                        end_path = true;
                    } else {
                        const Surface hit = make_surface(K, h, ro, rd, nullptr, nullptr);
                        if (bounce == 0 && s == 0) {
                            K.normal[idx * 3 + 0] = hit.normal.x;
                            K.normal[idx * 3 + 1] = hit.normal.y;
                            K.normal[idx * 3 + 2] = hit.normal.z;
                            K.depth[idx] = hit.t;
                            K.object_id[idx] = h.mesh;
                        }
                        const float4 m0 = K.materials[h.mesh * 6 + 0], m2 = K.materials[h.mesh * 6 + 2];
                        if (!hit.front_face) {
                            const f3 T_unit = mk3(max_(1e-6f, m0.x), max_(1e-6f, m0.y), max_(1e-6f, m0.z));
                            const f3 absorption = mk3(-det_log(T_unit.x), -det_log(T_unit.y), -det_log(T_unit.z));
                            throughput = throughput * beerLambert(absorption, hit.t);
                        }
                        if (m2.x > 0.0f || m2.y > 0.0f || m2.z > 0.0f) {
                            if (bounce == 0 || prev_was_specular)
                                acc = acc + throughput * mk3(m2.x, m2.y, m2.z);
                        }
                        f3 L = mk3(0.0f), light_scale = mk3(0.0f), shadow_o = mk3(0.0f);
                        float pdf_sample = 1.0f, shadow_tmax = 0.0f, light_att = 1.0f;
                        if (!ray_spec && K.n_lights > 0) {
                            float r = rng_uniform(rng);
                            r = min_(r, 0.99999994f);
                            const int light_index = (int)(r * (float)K.n_lights);
                            const LightRec light = load_light(K.lights, light_index);
                            const float pdf_pick = 1.0f / (float)K.n_lights;
                            float attenuation = 1.0f;
                            float light_dist = 1e30f;
                            const f3 light_radiance = light.color * light.intensity;
                            if (light.type == 1) {
                                L = -light.direction;
                                pdf_sample = pdf_pick;
                            } else {
                                const f3 toLight = light.position - hit.point;
                                const float light_dist_sq = dot(toLight, toLight);
                                light_dist = sqrt_ieee(light_dist_sq);
                                if (light.radius <= 0.0f) {
                                    L = toLight / light_dist;
                                    pdf_sample = pdf_pick;
                                } else {
                                    float sin_theta_max_sq = (light.radius * light.radius) / light_dist_sq;
                                    sin_theta_max_sq = min_(sin_theta_max_sq, 0.9999f);
                                    const float cos_theta_max = sqrt_ieee(1.0f - sin_theta_max_sq);
                                    L = sample_cone_direction(rng, toLight / light_dist, cos_theta_max);
                                    const float solid_angle = TWO_PI_F * (1.0f - cos_theta_max);
                                    pdf_sample = (solid_angle > 1e-6f) ? (pdf_pick / solid_angle) : pdf_pick;
                                }
                                attenuation = attenuate(light_dist, light.range);
                                if (light.type == 2) {
                                    const float theta = dot(L, -light.direction);
                                    const float epsilon = light.inner - light.outer;
                                    float spotIntensity;
                                    if (epsilon <= 1e-6f)
                                        spotIntensity = (theta >= light.outer) ? 1.0f : 0.0f;
                                    else
                                        spotIntensity = clampf((theta - light.outer) / epsilon, 0.0f, 1.0f);
                                    attenuation *= spotIntensity;
                                }
                            }
                            const f3 shadow_offset = dot(hit.normal, L) > 0.0f ? hit.normal * 1e-4f : -hit.normal * 1e-4f;
                            shadow_o = hit.point + shadow_offset;
                            shadow_tmax = light_dist - 1e-3f;
                            light_scale = light_radiance;
                            light_att = attenuation;
                            want_shadow = true;
                            ++n_shadow;
                        }
                        const Material mat = load_material(K.materials, h.mesh);
                        const f3 V = -rd;
                        if (want_shadow) {
                            // what an unoccluded light sample adds (path_logic.cuh:357-381); kept until the
                            // shadow ray has been traced
                            const f3 bsdf = evaluateBSDF<FULL>(hit, mat, L, V);
                            if (pdf_sample > 0.0f) {
                                f3 direct = bsdf * light_scale * light_att / pdf_sample;
                                direct = clamp_vector_soft(direct, 500.0f);
                                if (direct.x > 0.0f || direct.y > 0.0f || direct.z > 0.0f) {
                                    const float pdf_brdf = material_pdf<FULL>(hit, mat, V, L);
                                    const float wgt = mis_weight(pdf_sample, pdf_brdf);
                                    const f3 c = throughput * direct * wgt;
                                    W.pend[q] = c.x;
                                    W.pend[N + q] = c.y;
                                    W.pend[2 * N + q] = c.z;
                                    flags |= WF_PVALID;
                                }
                            }
                            if (flags & WF_PVALID) { // (a sample that adds nothing either way is counted, not walked)
                                W.sh[q] = shadow_o.x;
                                W.sh[N + q] = shadow_o.y;
                                W.sh[2 * N + q] = shadow_o.z;
                                W.sh[3 * N + q] = L.x;
                                W.sh[4 * N + q] = L.y;
                                W.sh[5 * N + q] = L.z;
                                W.sh[6 * N + q] = shadow_tmax;
                                flags |= WF_SHADOW;
                            } else {
                                ++n_zero;
                            }
                        }
                        f3 scatter_dir = mk3(0.0f), att = mk3(0.0f);
                        bool is_specular = false;
                        if (!material_scatter<FULL>(hit, mat, rd, rng, scatter_dir, att, is_specular)) {
                            end_path = true;
                        } else {
                            prev_was_specular = is_specular;
                            bool killed = false;
                            if (bounce >= 2) {
                                const float p = max_(0.05f, min_(0.95f, max_(throughput.x, max_(throughput.y, throughput.z))));
                                if (rng_uniform(rng) > p)
                                    killed = true;
                                else
                                    throughput = throughput / p;
                            }
                            if (killed) {
                                end_path = true;
                            } else {
                                throughput = throughput * att;
                                throughput = clamp_vector_soft(throughput, 50.0f);
                                const f3 off = hit.normal * 1e-4f;
                                ro = (dot(scatter_dir, hit.normal) > 0.0f) ? (hit.point + off) : (hit.point - off);
                                rd = scatter_dir;
                                ray_spec = is_specular;
                                ++bounce;
                                if (bounce >= K.max_depth)
                                    end_path = true;
                            }
                        }
                    }
                    if (end_path) {
                        if (want_shadow)
                            flags |= WF_ENDED; // finished once the light sample is resolved (next shade)
                        else
                            regen = true;
                    } else {
                        flags |= WF_EXT;
                    }
                }
            }
            bool done = false;
            if (regen) {
                if (!first) { // the sample is complete (scene_kernels.cuh:170-171)
                    acc = clamp_vector_soft(acc, 100.0f);
                    const f3 a = mk3(W.avg[q], W.avg[N + q], W.avg[2 * N + q]) + acc;
                    W.avg[q] = a.x;
                    W.avg[N + q] = a.y;
                    W.avg[2 * N + q] = a.z;
                    ++s;
                } else {
                    W.avg[q] = 0.0f;
                    W.avg[N + q] = 0.0f;
                    W.avg[2 * N + q] = 0.0f;
                }
                if (s >= K.spp) {
                    done = true;
                } else { // primary ray (scene_kernels.cuh:147-167, camera.cuh:156-205)
                    float tjx, tjy, bnx, bny;
                    taa_jitter(K.frame_count + s, tjx, tjy);
                    blue_noise_jitter(K.blue_noise, x, y, K.frame_count + s, bnx, bny);
                    const float jitter_x = tjx + (bnx - 0.5f) * 0.25f;
                    const float jitter_y = tjy + (bny - 0.5f) * 0.25f;
                    const float u = ((float)x + 0.5f + jitter_x) / (float)K.width;
                    const float v = 1.0f - ((float)y + 0.5f + jitter_y) / (float)K.height;
                    if (K.cam.lens_radius <= 0) {
                        const f3 dir = K.cam.llc + u * K.cam.horizontal + v * K.cam.vertical - K.cam.origin;
                        ro = K.cam.origin;
                        rd = normalize(dir);
                    } else {
                        f3 p;
                        do {
                            const float a = rng_uniform(rng);
                            const float b = rng_uniform(rng);
                            p = 2.0f * mk3(a, b, 0.0f) - mk3(1.0f, 1.0f, 0.0f);
                        } while (dot(p, p) >= 1.0f);
                        const f3 rdisk = K.cam.lens_radius * p;
                        const f3 offset = K.cam.u * rdisk.x + K.cam.v * rdisk.y;
                        const f3 dir = K.cam.llc + u * K.cam.horizontal + v * K.cam.vertical - K.cam.origin - offset;
                        ro = K.cam.origin + offset;
                        rd = normalize(dir);
                    }
                    ray_spec = true;
                    prev_was_specular = true;
                    throughput = mk3(1.0f);
                    acc = mk3(0.0f);
                    bounce = 0;
                    flags |= WF_EXT;
                }
            }
            K.rng[idx] = rng.d;
            K.rng[npix + idx] = rng.v0;
            K.rng[2 * npix + idx] = rng.v1;
            K.rng[3 * npix + idx] = rng.v2;
            K.rng[4 * npix + idx] = rng.v3;
            K.rng[5 * npix + idx] = rng.v4;
            if (done) {
                n_paths += (uint32_t)K.spp;
                const f3 out = mk3(W.avg[q], W.avg[N + q], W.avg[2 * N + q]) / (float)K.spp;
                K.accum[idx * 3 + 0] = out.x;
                K.accum[idx * 3 + 1] = out.y;
                K.accum[idx * 3 + 2] = out.z;
                if (K.rgb8) {
                    unsigned char r8, g8, b8;
                    tonemap_pixel(out, r8, g8, b8);
                    const size_t o = ((size_t)rgb8_row(K, yl) * K.width + x) * 3;
                    K.rgb8[o + 0] = r8;
                    K.rgb8[o + 1] = g8;
                    K.rgb8[o + 2] = b8;
                }
                W.st[q] = WF_DONE;
            } else {
                W.thr[q] = throughput.x;
                W.thr[N + q] = throughput.y;
                W.thr[2 * N + q] = throughput.z;
                W.acc[q] = acc.x;
                W.acc[N + q] = acc.y;
                W.acc[2 * N + q] = acc.z;
                if (flags & WF_EXT) {
                    W.ray[q] = ro.x;
                    W.ray[N + q] = ro.y;
                    W.ray[2 * N + q] = ro.z;
                    W.ray[3 * N + q] = rd.x;
                    W.ray[4 * N + q] = rd.y;
                    W.ray[5 * N + q] = rd.z;
                }
                W.st[q] = (uint32_t)s | ((uint32_t)bounce << 8) | (ray_spec ? WF_SPEC : 0u) |
                          (prev_was_specular ? WF_PREV_SPEC : 0u) | flags;
                alive_after = true;
            }
        }
    }
}

// Active-path sorting (option "wf_sort"; north_star: "wavefront ballot/prefix-sum for ray compaction and active-path sorting";
// path_logic.cuh:490-780 is where a wave of unsorted paths diverges: transmissive / clear-coated / iridescent / sheen / plain
// branches of material_scatter, evaluateBSDF and material_pdf, six importance_sample_ggx sites).  The paths are one frame-wide pool
// in tile order; a workgroup takes WF_SORT_G x 256 consecutive ones, bins them by class in LDS -- counting sort: LDS histogram,
// a wave's prefix sum over the classes, LDS cursors -- and its waves then shade runs of ONE class: finished and idle paths
// first (those waves leave at once: compaction), then regenerations, misses, and the hits by {mesh = material, the ray's
// specular flag, bounce >= 2 (roulette)}.  Which thread shades a path changes nothing in the path: same bits.
constexpr int WF_CLASSES = 3 + 64 * 4; // (WF_SORT_G, the kernel's third template argument: 256-path groups a workgroup sorts together, 0 = no sorting)
PT_DEV int wf_class(const WfParams &W, int q) {
    if (q >= W.n_items)
        return 0;
    const uint32_t st = W.st[q];
    if (st & WF_DONE)
        return 0;
    if ((st & WF_ENDED) || !(st & WF_EXT))
        return 1; // resolves its light sample and starts the next sample
    const int mesh = __float_as_int(W.hit[q].z);
    if (mesh < 0)
        return 2; // sky
    return 3 + (mesh & 63) * 4 + ((st & WF_SPEC) ? 1 : 0) + ((((st >> 8) & 0xffu) >= 2u) ? 2 : 0);
}

template <bool FULL, int WF_SORT_G = 0> __global__ __launch_bounds__(256) void wf_shade_kernel(const KParams K, const WfParams W) {
    constexpr bool SORT = WF_SORT_G > 0;
    if (W.iter > 0 && W.live[W.iter - 1] == 0u)
        return;
    const int lane = threadIdx.x & 63;
    WfCounts cn;
    if (SORT) {
        __shared__ int hist[WF_CLASSES + 1], order[256 * (SORT ? WF_SORT_G : 1)];
        const int seg0 = blockIdx.x * 256 * WF_SORT_G;
        for (int i = threadIdx.x; i <= WF_CLASSES; i += 256)
            hist[i] = 0;
        __syncthreads();
        int cls[SORT ? WF_SORT_G : 1];
#pragma unroll
        for (int g = 0; g < WF_SORT_G; ++g) {
            cls[g] = wf_class(W, seg0 + g * 256 + (int)threadIdx.x);
            atomicAdd(&hist[cls[g]], 1);
        }
        __syncthreads();
        if (threadIdx.x < 64) { // exclusive prefix sum over the classes, 64 at a time (one wave)
            int carry = 0;
            for (int c0 = 0; c0 < WF_CLASSES; c0 += 64) {
                const int c = c0 + lane;
                const int v = c < WF_CLASSES ? hist[c] : 0;
                int inc = v;
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(inc, off);
                    if (lane >= off)
                        inc += t;
                }
                if (c < WF_CLASSES)
                    hist[c] = carry + inc - v;
                carry += __shfl(inc, 63);
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < WF_SORT_G; ++g)
            order[atomicAdd(&hist[cls[g]], 1)] = seg0 + g * 256 + (int)threadIdx.x;
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < WF_SORT_G; ++g)
            wf_shade_path<FULL>(K, W, order[g * 256 + threadIdx.x], cn);
    } else {
        wf_shade_path<FULL>(K, W, blockIdx.x * 256 + threadIdx.x, cn);
    }
    const uint32_t n_ext = cn.n_ext, n_shadow = cn.n_shadow, n_paths = cn.n_paths, n_zero = cn.n_zero;
    if (__builtin_amdgcn_ballot_w64(cn.alive_after) && lane == 0)
        W.live[W.iter] = 1u; // same value from every writer
    if (K.counters) {
        uint32_t a = n_ext, b = n_shadow, c = n_paths, z = n_zero;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
            c += __shfl_xor(c, off);
            z += __shfl_xor(z, off);
        }
        if (lane == 0 && (a | b | c)) { // one slot per wave of this grid; launches are ordered
            unsigned long long *w = K.counters + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * COUNTER_WORDS;
            w[0] += (unsigned long long)a;
            w[1] += (unsigned long long)b;
            w[2] += (unsigned long long)c;
            w[3] += (unsigned long long)z;
        }
    }
}

} // namespace pt
