// pt_async.hip.h -- path_trace_async_kernel<FULL>: the render loop as a persistent megakernel whose
// LANES run asynchronously (scenes whose BLASes are real trees behind a single-leaf TLAS).
//
// Measured on the showcase scene (PT_TRAV_STATS, tools/trav_stats.py): in path_trace_kernel a
// traversal phase lasts as long as the longest of the wave's <= 64 rays -- 39 node iterations for
// 8.3 nodes per ray: 17 % of the lanes busy in the closest-hit loops, 6 % in the shadow loops.
// Splitting the frame into trace / shade launches (pt_wavefront.hip.h) fills the lanes but pays the
// longest ray of the FRAME at every one of its ~20 launches (0.2-0.6 ms each): slower overall.
// Here nothing waits for anything but itself:
//
//   * every lane is a little state machine   PIXEL -> TRACE -> HIT -> [TRACE -> SHADOW] -> TRACE ...
//     A lane whose ray is finished shades its own hit (emission, light sample -> its own shadow ray;
//     BSDF sample, roulette -> its next ray; next sample; next pixel) while the other lanes keep
//     walking their trees.  The shading block runs when shade_min lanes wait for it (or no lane is
//     tracing), so its cost is shared; the traversal loops see a full wave almost all the time.
//   * a lane walks its ray through the meshes whose root box it hits, in TLAS order, exactly like one
//     thread of the reference (bvh_trace_tlas / bvh_any_hit_tlas, intersection.cuh:438-605); shadow
//     and extension rays share one code path (any-hit order does not matter, E4).
//   * pixels come from a frame-wide pool: one atomic per 8x8 tile refills the wave's LDS ring, lanes
//     take pixels from the ring one by one, so a wave never idles behind its slowest pixel and the grid
//     is just big enough to fill the chip (persistent waves).
//
// Per pixel the random numbers are drawn in the reference's order and every float is combined in the
// reference's order (the code between the traces is phases [A], [C], [E] of path_trace_kernel), so
// the bits are the megakernel's and the oracle's: tests/test_async_gpu.py.
#pragma once
#include "pt_wavefront.hip.h"

namespace pt {

struct AsyncParams {
    uint32_t *cursor; // next tile of the frame-wide pool (zeroed before the launch)
    int n_tiles;
    int shade_min; // lanes that must wait for the shading block before it runs while others still trace
    int leaf_min;  // the node loop stops once this many lanes wait at a leaf (64 = classic while-while: all of them)
};
constexpr int AS_RING = 128;

template <bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_WAVES_PER_EU, 8))) void path_trace_async_kernel(
    const KParams K, const AsyncParams A) {
    extern __shared__ uint2 lds_raw[];
    const int lane = threadIdx.x;
    LdsStack stk{lds_raw + lane};
    uint32_t *ring = (uint32_t *)(lds_raw + K.stack_entries * 64);
    unsigned long long *lkey = (unsigned long long *)(ring + AS_RING); // 64 merge keys of the leaf phase
    unsigned char *owner = (unsigned char *)(lkey + 64);               // 64 * 17 rounded up: lane of each test
    int ring_head = 0, ring_n = 0; // wave-uniform
    bool pool_empty = false;
    const int2 tl = K.tlas_leaves[~K.tlas_root_ref];
    const int M = tl.y;
    const size_t npix = K.rng_plane;

    enum { ST_PIXEL = 0, ST_TRACE = 1, ST_HIT = 2, ST_SHADOW = 3, ST_RETIRED = 4 };
    int stage = ST_PIXEL;
    // pixel and path
    int x = 0, yl = 0, s = 0, bounce = 0;
    size_t idx = 0;
    bool ray_spec = true, prev_was_specular = true;
    Rng rng = {0, 0, 0, 0, 0, 0};
    f3 throughput = mk3(1.0f), acc = mk3(0.0f), avg_color = mk3(0.0f);
    f3 ro = mk3(0.0f), rd = mk3(0.0f); // ro: origin of the ray being traced (extension OR shadow); rd: path direction
    // what the second half of the shading needs of the first, across the shadow ray
    Surface hit;
    hit.point = hit.normal = mk3(0.0f);
    hit.t = 0.0f;
    hit.front_face = true;
    f3 L = mk3(0.0f), light_scale = mk3(0.0f);
    float pdf_sample = 1.0f, light_att = 1.0f;
    int hmesh = -1;
    // traversal
    bool active = false, anyq = false, xf = false, found = false;
    int cur = 0, sp = 0, sb = -1, mi = 0, gmesh = -1, gslot = -1;
    unsigned long long mask = 0ull;
    float ds = 1.0f, tb = T_FAR, tm = T_FAR, gt = T_FAR, gtl = T_FAR;
    RayO pr = make_ray(mk3(0.0f), mk3(0.0f, 0.0f, 1.0f));
    uint32_t n_ext = 0, n_shadow = 0, n_paths = 0;
    // PT_TRAV_STATS: 0 main-loop iterations, 1 shading-block runs, 2 lanes shading in them, 3 between-mesh
    // block runs, 4 lanes in them, 5 node wave-iterations, 6 node lane-steps, 7 leaf phases (unused here),
    // [8] triangle wave-iterations, [9] triangle lane-tests, [10] lanes tracing summed over main-loop iterations
    TravStats ts;
#ifdef PT_TRAV_STATS
    uint32_t ts2[3] = {0, 0, 0};
#endif

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

    for (;;) {
        TS_WAVE(0);
        // ---- keep the wave's pixel ring stocked: one atomic per 8x8 tile
        if (!pool_empty && ring_n < 64) {
            uint32_t t = 0u;
            if (lane == 0)
                t = atomicAdd(A.cursor, 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if ((int)t < A.n_tiles) {
                ring[(ring_head + ring_n + lane) & (AS_RING - 1)] = t * 64u + (uint32_t)lane;
                ring_n += 64;
            } else {
                pool_empty = true;
            }
            wave_lds_order();
        }

        // ---- shading block: lanes whose ray is finished, lanes without a pixel
        const unsigned long long want = __builtin_amdgcn_ballot_w64(stage != ST_TRACE && stage != ST_RETIRED);
        const unsigned long long tracing = __builtin_amdgcn_ballot_w64(stage == ST_TRACE);
        if (want && (__builtin_popcountll(want) >= A.shade_min || !tracing)) {
            bool newray = false, need_regen = false;
            TS_WAVE(1);
            if (stage == ST_HIT || stage == ST_SHADOW) {
                TS_LANE(2);
                bool end_path = false, run_e = (stage == ST_SHADOW), want_shadow = (stage == ST_SHADOW);
                if (stage == ST_HIT) { // ---- [C] (scene_kernels.cuh:181-193, path_logic.cuh:795-840, 305-382)
                    ++n_ext;
                    Hit h;
                    h.t = gt;
                    h.t_local = gtl;
                    h.u = h.v = 0.0f;
                    h.mesh = gmesh;
                    h.slot = gslot;
                    hmesh = gmesh;
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
                        run_e = true;
                        hit = make_surface(K, h, ro, rd, nullptr, nullptr);
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
                            light_scale = light_radiance;
                            light_att = attenuation;
                            ++n_shadow;
                            // this lane's next ray is its shadow ray; [E] follows when that is finished
                            ro = hit.point + shadow_offset;
                            tm = light_dist - 1e-3f;
                            anyq = true;
                            newray = true;
                            run_e = false;
                        }
                    }
                }
                if (run_e) { // ---- [E] (path_logic.cuh:357-381, 843-896)
                    const Material mat = load_material(K.materials, hmesh);
                    const f3 V = -rd;
                    if (want_shadow && !found) {
                        const f3 bsdf = evaluateBSDF<FULL>(hit, mat, L, V);
                        if (pdf_sample > 0.0f) {
                            f3 direct = bsdf * light_scale * light_att / pdf_sample;
                            direct = clamp_vector_soft(direct, 500.0f);
                            if (direct.x > 0.0f || direct.y > 0.0f || direct.z > 0.0f) {
                                const float pdf_brdf = material_pdf<FULL>(hit, mat, V, L);
                                const float wgt = mis_weight(pdf_sample, pdf_brdf);
                                acc = acc + throughput * direct * wgt;
                            }
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
                    if (!end_path) {
                        anyq = false;
                        newray = true;
                    }
                }
                if (end_path) { // the sample is complete (scene_kernels.cuh:170-171)
                    acc = clamp_vector_soft(acc, 100.0f);
                    avg_color = avg_color + acc;
                    ++s;
                    if (s < K.spp) {
                        need_regen = true;
                    } else { // the pixel is complete (scene_kernels.cuh:173-193, tonemap_kernel scene.cuh:2004-2047)
                        K.rng[idx] = rng.d;
                        K.rng[npix + idx] = rng.v0;
                        K.rng[2 * npix + idx] = rng.v1;
                        K.rng[3 * npix + idx] = rng.v2;
                        K.rng[4 * npix + idx] = rng.v3;
                        K.rng[5 * npix + idx] = rng.v4;
                        const f3 out = avg_color / (float)K.spp;
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
                        n_paths += (uint32_t)K.spp;
                        stage = ST_PIXEL;
                    }
                }
            }
            // ---- lanes without a pixel take the next ones of the ring (rank by ballot / prefix sum)
            {
                const unsigned long long takers = __builtin_amdgcn_ballot_w64(stage == ST_PIXEL);
                const int n_takers = __builtin_popcountll(takers);
                const int n_take = n_takers < ring_n ? n_takers : ring_n;
                if (stage == ST_PIXEL) {
                    const int rank = lane_prefix(takers);
                    if (rank < n_take) {
                        const uint32_t q = ring[(ring_head + rank) & (AS_RING - 1)];
                        const int tile = (int)(q >> 6), l = (int)(q & 63u);
                        const int tx = tile % K.tiles_x, ty = tile / K.tiles_x;
                        x = tx * 8 + (l & 7);
                        yl = ty * 8 + (l >> 3);
                        if (x < K.width && yl < K.rows) { // (a tile on the frame's edge has positions outside it)
                            idx = (size_t)yl * K.width + x;
                            rng.d = K.rng[idx];
                            rng.v0 = K.rng[npix + idx];
                            rng.v1 = K.rng[2 * npix + idx];
                            rng.v2 = K.rng[3 * npix + idx];
                            rng.v3 = K.rng[4 * npix + idx];
                            rng.v4 = K.rng[5 * npix + idx];
                            avg_color = mk3(0.0f);
                            s = 0;
                            need_regen = true;
                        }
                    } else if (pool_empty) {
                        stage = ST_RETIRED;
                    }
                }
                ring_head = (ring_head + n_take) & (AS_RING - 1);
                ring_n -= n_take;
            }
            if (need_regen) { // ---- [A] primary ray (scene_kernels.cuh:147-167, camera.cuh:156-205)
                const int y = global_row(yl, K.y0, K.il_period, K.il_phase);
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
                anyq = false;
                newray = true;
            }
            if (newray) { // ---- root boxes: the TLAS node, then every mesh of its leaf
                if (!anyq)
                    tm = T_FAR;
                const RayO w = make_ray(ro, anyq ? L : rd);
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
                        float sc;
                        const RayO lr = local_ray(K, m, w, sc);
                        hb = in && slab(mh.bmin, mh.bmax, lr, anyq ? tm * sc : T_FAR, tE);
                    } else {
                        hb = in && slab(mh.bmin, mh.bmax, w, tm, tE);
                    }
                    mask |= hb ? (1ull << i) : 0ull;
                }
                pr = w;
                xf = false;
                active = false;
                found = false;
                sb = -1;
                sp = 0;
                gt = gtl = T_FAR;
                gmesh = gslot = -1;
                stage = ST_TRACE;
            }
        }
        if (!__builtin_amdgcn_ballot_w64(stage != ST_RETIRED))
            break;

        // ---- a tracing lane between meshes: keep the mesh's closest hit, then the next mesh or the result
#ifdef PT_TRAV_STATS
        if (stage == ST_TRACE)
            ++ts2[2];
#endif
        if (stage == ST_TRACE && !active) {
            TS_WAVE(3);
            TS_LANE(4);
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
                stage = anyq ? ST_SHADOW : ST_HIT;
            } else {
                const int i = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                mi = K.tlas_mesh_ids[tl.x + i];
                const float4 r0 = K.mesh_recs[mi * MESH_REC_F4], r1 = K.mesh_recs[mi * MESH_REC_F4 + 1];
                const bool nxf = (__float_as_int(r1.w) & 1) != 0;
                if (nxf || xf) { // entering or leaving a mesh's local space
                    const RayO w = make_ray(ro, anyq ? L : rd);
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
        // ---- inner nodes.  Lanes that have reached a leaf wait for the others, but not for all of them:
        // a lane needs ~2 node steps to its next leaf while the slowest of 64 needs ~12, so the loop
        // ends as soon as leaf_min lanes are waiting (they are served, pop, and rejoin the descent)
        for (;;) { // (wave-uniform loop, predicated step: the lanes at a leaf take part in the ballots)
            const bool innode = active && cur >= 0;
            if (!__builtin_amdgcn_ballot_w64(innode))
                break;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(active && cur < 0)) >= A.leaf_min)
                break;
            if (innode) {
                TS_WAVE(5);
                TS_LANE(6);
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
        }
        // ---- leaf phase as (lane, triangle) pairs (see closest_hit_pairs_dyn in pt_render.hip.h): the tests of all
        // lanes waiting at a leaf form one list, 64 per iteration; ray, limit and leaf start come out of the
        // owner lane's registers (ds_bpermute); merge by LDS 64-bit min {t bits, index} -- or a flag for shadow rays
        {
            const bool atleaf = active && cur < 0;
            if (__builtin_amdgcn_ballot_w64(atleaf)) {
                int cnt = 0, first = 0;
                if (atleaf) {
                    const int2 lf = K.leaves[~cur];
                    first = lf.x;
                    cnt = lf.y;
                }
                int incl = cnt;
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(incl, off);
                    if (lane >= off)
                        incl += v;
                }
                const int start = incl - cnt;
                const int T = __builtin_amdgcn_readlane(incl, 63);
                if (atleaf) {
                    lkey[lane] = ~0ull;
                    for (int i = 0; i < cnt; ++i)
                        owner[start + i] = (unsigned char)lane;
                }
                wave_lds_order();
                for (int j0 = 0; j0 < T; j0 += 64) {
                    const int j = j0 + lane;
                    const int o = owner[j < T ? j : 0];
                    const int i = j - __shfl(start, o);
                    const int slot = __shfl(first, o) + i;
                    const float tl = __shfl(tb, o);
                    RayO tr;
                    tr.o = mk3(__shfl(pr.o.x, o), __shfl(pr.o.y, o), __shfl(pr.o.z, o));
                    tr.d = mk3(__shfl(pr.d.x, o), __shfl(pr.d.y, o), __shfl(pr.d.z, o));
                    if (j < T) {
#ifdef PT_TRAV_STATS
                        if (lane == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                            ++ts2[0];
                        ++ts2[1];
#endif
                        const float4 *tp = K.tris + (size_t)slot * 3;
                        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                        float t, u, v;
                        if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), tr, tl, t, u, v))
                            __hip_atomic_fetch_min(&lkey[o], ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)i,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                wave_lds_order();
                if (atleaf) {
                    const unsigned long long key = lkey[lane];
                    if (key != ~0ull) {
                        if (anyq) {
                            found = true;
                        } else {
                            tb = __uint_as_float((uint32_t)(key >> 32));
                            sb = first + (int)(uint32_t)key;
                        }
                    }
                    if (found)
                        active = false;
                    else
                        pop();
                }
            }
        }
    }

    ts.flush(0, lane);
#ifdef PT_TRAV_STATS
    for (int i = 0; i < 3; ++i) {
        unsigned a = ts2[i];
        for (int off = 32; off > 0; off >>= 1)
            a += __shfl_xor(a, off);
        if (lane == 0)
            atomicAdd(&g_trav_stats[8 + i], (unsigned long long)a);
    }
#endif
    if (K.counters) {
        uint32_t a = n_ext, b = n_shadow, c = n_paths;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
            c += __shfl_xor(c, off);
        }
        if (lane == 0) { // one slot per workgroup (the grid never exceeds the number of tiles)
            unsigned long long *w = K.counters + (size_t)blockIdx.x * COUNTER_WORDS; // (this shape walks every shadow ray: word 3 stays 0)
            w[0] += (unsigned long long)a;
            w[1] += (unsigned long long)b;
            w[2] += (unsigned long long)c;
        }
    }
}

} // namespace pt
