// pt_kernels.hip.h -- gfx950 kernels of the path tracer.
//
//   path_trace_kernel<GEOM,FULL>  the per-pixel render loop: primary-ray generation,
//        two-level BVH closest hit, shading + next-event estimation (shadow any-hit),
//        BSDF sampling, Russian roulette, sample accumulation, G-buffer, RNG
//        write-back and the tonemap, in ONE launch.
//        (reference: path_trace_kernel scene_kernels.cuh:122-194 -> tracePath
//        path_logic.cuh:782-899 -> traceRay intersection.cuh:526-605, and
//        tonemap_kernel scene.cuh:2004-2047)
//   xorwow_init_kernel            init_curand_kernel (scene_kernels.cuh:26-35)
//   trace_rays_kernel             trace_single_ray_kernel (scene_kernels.cuh:38-49), batched
//
// Execution model (MI355X-first, not the reference's 8x8 CUDA block of 2 warps):
//   * one 64-lane wavefront = one 8x8 pixel tile = one workgroup; lanes keep their
//     pixel for the whole frame because the per-pixel XORWOW stream is strictly
//     sequential across samples and bounces.
//   * persistent lanes with path regeneration: a lane whose path ends starts its
//     next sample immediately instead of idling until the slowest lane of the wave
//     finishes the bounce loop.
//   * scene data is re-laid-out at upload: child-pair 64-byte BVH nodes (one fetch
//     per inner node gives both child boxes), 48-byte triangle packets
//     {v0|face, e1, e2} in leaf order (no index indirections), 96-byte material and
//     64-byte light records, all float4-aligned.
//   * wave-uniform data (mesh records, single-leaf triangle packets, lights,
//     materials of a uniform mesh) is addressed through readfirstlane'd indices so
//     the compiler fetches it with scalar loads into SGPRs.
//   * per-lane traversal stacks live in LDS, [depth][lane] layout (ds_*_b64,
//     conflict-free), sized by the scene's real BVH depth.
//
// Equivalences used (each keeps results bit-identical to the literal algorithm;
// proofs in DESIGN.md "Traversal equivalences"):
//   E1 the reference re-tests a node's own box when it becomes current; for the
//      near child that test repeats the one just made, for a popped far child it
//      reduces to `entry < best.t`, so the stack carries the entry distance.
//   E2 the slab test's early-out after the y slab never changes the result.
//   E3 one running closest hit with strict `<` over (mesh order, leaf order) equals
//      per-mesh closest followed by strict `<` across meshes, for untransformed
//      meshes (transformed ones keep a separate local-space best).
//   E4 any-hit is order independent as long as the same boxes gate the same
//      triangles; BVHs deeper than 23 levels are rejected at upload, so the
//      reference's 24-entry stack-overflow drop can never trigger.
#pragma once
#include "pt_device.hip.h"

namespace pt {

struct Camera {
    f3 origin, llc, horizontal, vertical, u, v, w;
    float lens_radius;
};

// Everything the kernels read, by value in the kernarg segment.
struct KParams {
    // scene arena (device pointers)
    const float4 *mesh_recs;  // 12 float4 per mesh
    const float4 *nodes;      // 4 float4 per inner node (child pair)
    const float4 *nodes2;     // NODE2_F4 float4 per inner node: the node, its left child's node, its right child's node (two levels
                              // per fetch: expand_nodes_kernel derives them from `nodes` after every upload / refit / rebuild)
    const int2 *leaves;       // {first tri slot, count}
    const float4 *tris;       // 3 float4 per leaf slot: {v0, e1, e2}; the three w hold the geometric normal (tri_normals_kernel)
    const int4 *slot_face;    // per leaf slot: global vertex indices + face index
    const float4 *tlas_nodes; // child-pair nodes over meshes
    const int2 *tlas_leaves;  // {first index into tlas_mesh_ids, count}
    const int *tlas_mesh_ids;
    const float4 *tlas_heads; // PMODE 3: per TLAS index (leaf order) TLAS_HEAD_F4 float4, see gather_tlas_heads_kernel
    float inst_c2;            // PMODE 3: first-pass boxes of instances grow by inst_c2 * (|o.x| + |o.y| + |o.z|) per ray
    const float4 *materials; // 6 float4 per mesh
    const float4 *lights;    // 4 float4 per light
    const float2 *blue_noise; // 64*64
    const float4 *tlas_root_box; // {bmin, bmax} of TLAS node 0, in device memory so ptrt_refit can move it
    int tlas_root_ref;           // >=0 inner node, <0 ~leaf
    int n_meshes, n_lights;
    int stack_entries; // LDS stack depth per lane (BLAS)
    // pair-compacted tracing (pt_render.hip.h), valid when every BLAS is a single leaf
    int pair_meshes;    // meshes in the (single) TLAS leaf
    int pair_tri_slots; // triangle packets staged in LDS (all leaf slots, 0..n)
    int pair_max_leaf;  // largest leaf
    int tlas_max_leaf;  // PMODE 3: most meshes in one TLAS leaf
    int tlas_depth;     // PMODE 3: TLAS stack entries per lane
    int tlas_any_rounds; // PMODE 3 shadow rays: 1 = one TLAS leaf per ray and fill (more than 1024 meshes, or option tlas_rounds)
    int pair_split;     // PMODE 1: a batch that does not fill the wave may give each pair several lanes
    int steal;          // PMODE 2 any-hit: 0 off; n > 0: idle lanes steal subtrees, node loop yields every n steps
    int csteal;         // PMODE 2 closest hit: 0 off; n > 0: verified subtree stealing (run_closest_queue), node loop yields every n steps
    int csteal_min;     // ... node steps a walk must have taken before its stack may be stolen from
    int csteal_follow;  // ... 1: a thief keeps taking its victim's limit while that walk lasts
    int csteal_leaf_min; // ... lanes waiting at a leaf that end the node loop of a stealing closest-hit phase (leaf_min of the others)
    int leaf_min;       // PMODE 2: lanes waiting at a leaf that end the node loop (64 = all of them)
    int leaf_pairs;     // PMODE 2: leaf phase as compacted (lane, triangle) pairs
    int fetch_min;      // PMODE 2: idle lanes before the wave refills from the pair list (0 = static 64-pair batches)
    int n_nodes;        // inner nodes in the arena (the LDS-staged variant clamps its copies to them)
    int top_off;        // WG = 4 variant, A/B: do not read the staged nodes (isolates the cost of the larger workgroups)
    int n_tiles;        // 8x8-pixel tiles of this launch (the 4-wave variant's last workgroup may own fewer than 4)
    int lds_extra;      // PMODE 1: byte offset in the workgroup's LDS of the staged shading inputs its waves share: the jitter
                        // table (16 float2), then what lds_flags names
    int lds_flags;      // 0 = nothing staged; bit 2: the jitter table is; bit 3: and per wave the lanes' blue-noise values; bit 0: the light records follow the table (n_lights <= LDS_LIGHTS); bit 1: then the
                        // material records of the leaf's meshes, by mesh ORDER
    int pair_cap;       // PMODE 4: entries the LDS pair list holds (a multiple of 64, >= 64 * pair_meshes + 64)
    int lds_wave, lds_wave_bytes; // PMODE 1: byte offset of the first wave's own lists in the workgroup's LDS, and their stride
    // frame
    Camera cam;
    f3 sky_top, sky_bottom;
    int use_sky;
    const float4 *env; // equirectangular environment map (Scene::loadHDRI) or NULL
    int env_w, env_h;
    int width, height; // full frame
    int y0, rows;      // tile
    int il_period, il_phase; // > 1: the context owns the 8-row strips `phase, phase + period, ...` of the frame (global_row)
    int tiles_x;
    int split_n, split_i; // the frame's tile ROWS are dealt to split_n concurrent launches (0 / 1: one launch); this one takes rows
                          // split_i, split_i + split_n, ...
    int spp, max_depth, frame_count;
    int sample_sync; // 1: the lanes of a wave start their samples together (path_trace_kernel [A])
    int tile_run;    // option "tile_run": workgroup -> tile map of the one-tile kernels, see path_trace_kernel (0: workgroup k renders tile k)
    // buffers (tile-sized)
    uint32_t *rng;    // 6 planes of rng_plane words (the context's rows*width; larger than the frame at a reduced render size)
    size_t rng_plane;
    float *accum, *normal, *depth;
    int *object_id;
    unsigned char *rgb8;
    int rgb8_frame; // 0: rgb8 is this context's own image (its rows, bottom-up); 1: rgb8 is the whole W x H frame (bottom-up) and
                    // the context writes its rows where they belong in it (ptrt_render, PTRT_OUT_DEVICE_FRAME)
    int ticket_tiles;    // lane-refill variant: consecutive tiles per ticket (>= 1)
    unsigned int *queue; // lane-refill variant: {next ticket of the launch's tile queue, waves that have left}; zero between launches
    unsigned long long *counters; // COUNTER_WORDS per slot: {extension, shadow, paths, zero-valued light samples} or nullptr
};
constexpr int COUNTER_WORDS = 4;

// frame row of a context's local row: contiguous rows from y0, or every il_period-th 8-row strip from strip il_phase
PT_DEV int global_row(int yl, int y0, int il_period, int il_phase) {
    return il_period > 1 ? ((((yl >> 3) * il_period + il_phase) << 3) | (yl & 7)) : y0 + yl;
}

// byte row of the RGB8 target that the context's local row yl goes to (both images are bottom-up)
PT_DEV int rgb8_row(const KParams &K, int yl) {
    return K.rgb8_frame ? K.height - 1 - global_row(yl, K.y0, K.il_period, K.il_phase) : K.rows - 1 - yl;
}

constexpr int MESH_REC_F4 = 12;
constexpr int NODE2_F4 = 12;
constexpr int TOP_LEVELS = 3;                    // BLAS levels numbered in level order (ptrt_capi.hip convert_tree)
constexpr int TOP_NODES = (1 << TOP_LEVELS) - 1;  // ... = the first 7 inner nodes of a tree
constexpr int TLAS_HEAD_F4 = 7; // {first-pass box, root}, {.., flags}, the three rows of the inverse matrix, and for an instance its LOCAL box
constexpr float T_FAR = 1e30f;

PT_DEV f3 tlas_bmin(const KParams &K) {
    const float4 a = K.tlas_root_box[0];
    return mk3(a.x, a.y, a.z);
}
PT_DEV f3 tlas_bmax(const KParams &K) {
    const float4 a = K.tlas_root_box[1];
    return mk3(a.x, a.y, a.z);
}

struct RayO { // RayOptimized, intersection.cuh:39-88
    f3 o, d, inv;
    bool sx, sy, sz;
};
PT_DEV RayO make_ray(f3 o, f3 d) {
    RayO r;
    r.o = o;
    r.d = d;
    // 1/d through rcp_ieee (bit-identical to the division for every input, tests/test_misc_gpu.py): a ray is set up
    // per (ray, mesh) pair and per instance root test, and three compiler divisions were a fifth of that
    const bool bx = __builtin_fabsf(d.x) > 1e-8f, by = __builtin_fabsf(d.y) > 1e-8f, bz = __builtin_fabsf(d.z) > 1e-8f;
    r.inv.x = bx ? rcp_ieee(bx ? d.x : 1.0f) : ((d.x >= 0) ? 1e30f : -1e30f);
    r.inv.y = by ? rcp_ieee(by ? d.y : 1.0f) : ((d.y >= 0) ? 1e30f : -1e30f);
    r.inv.z = bz ? rcp_ieee(bz ? d.z : 1.0f) : ((d.z >= 0) ? 1e30f : -1e30f);
    r.sx = r.inv.x < 0;
    r.sy = r.inv.y < 0;
    r.sz = r.inv.z < 0;
    return r;
}

// aabb_hit_fast / aabb_hit_fast_t (intersection.cuh:136-216) in one branch-free form (E2).
// Hardware min/max are safe here: operands are never NaN and the sign of a zero
// only ever feeds comparisons.
PT_DEV bool slab(f3 bmin, f3 bmax, const RayO &r, float tMax, float &tEntry) {
    const float ax = (bmin.x - r.o.x) * r.inv.x, bx = (bmax.x - r.o.x) * r.inv.x;
    const float ay = (bmin.y - r.o.y) * r.inv.y, by = (bmax.y - r.o.y) * r.inv.y;
    const float az = (bmin.z - r.o.z) * r.inv.z, bz = (bmax.z - r.o.z) * r.inv.z;
    const float t0x = r.sx ? bx : ax, t1x = r.sx ? ax : bx;
    const float t0y = r.sy ? by : ay, t1y = r.sy ? ay : by;
    const float t0z = r.sz ? bz : az, t1z = r.sz ? az : bz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(t0x, t0y), t0z);
    const float tmax = __builtin_fminf(__builtin_fminf(t1x, t1y), t1z);
    tEntry = __builtin_fmaxf(tmin, 0.0f);
    return (tmax >= 0.0f) && (tmin <= tmax) && (tmin < tMax);
}

// triangle_intersect_fast (intersection.cuh:219-255) on a pre-differenced packet, plus
// the caller's `t > 1e-5f` acceptance (intersection.cuh:329,379), branch-free.
PT_DEV bool tri_test(f3 v0, f3 e1, f3 e2, const RayO &r, float tMax, float &t_out, float &u_out, float &v_out) {
    const f3 h = cross(r.d, e2);
    const float a = dot(e1, h);
    const float f = rcp_ieee(a);
    const f3 s = r.o - v0;
    const float u = f * dot(s, h);
    const f3 q = cross(s, e1);
    const float v = f * dot(r.d, q);
    const float t = f * dot(e2, q);
    const bool ok = !(__builtin_fabsf(a) < 1e-6f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) &&
                    (t > 1e-6f && t < tMax) && (t > 1e-5f);
    t_out = t;
    u_out = u;
    v_out = v;
    return ok;
}

PT_DEV f3 xform_point(const float4 r0, const float4 r1, const float4 r2, f3 p) { // intersection.cuh:258-263
    return mk3(r0.x * p.x + r0.y * p.y + r0.z * p.z + r0.w, r1.x * p.x + r1.y * p.y + r1.z * p.z + r1.w,
               r2.x * p.x + r2.y * p.y + r2.z * p.z + r2.w);
}
PT_DEV f3 xform_dir(const float4 r0, const float4 r1, const float4 r2, f3 d) { // intersection.cuh:266-271
    return mk3(r0.x * d.x + r0.y * d.y + r0.z * d.z, r1.x * d.x + r1.y * d.y + r1.z * d.z,
               r2.x * d.x + r2.y * d.y + r2.z * d.z);
}

struct Hit {
    float t;       // world-space distance (T_FAR = miss)
    float t_local; // distance along the mesh-local ray (== t for untransformed meshes)
    float u, v;
    int mesh; // -1 = miss
    int slot; // global leaf slot of the triangle
};

// LDS traversal stack of one wave: [entry][lane] of {ref, entry distance}
struct LdsStack {
    uint2 *base; // + lane
    PT_DEV void push(int sp, int ref, float t) { base[sp * 64] = make_uint2((uint32_t)ref, __float_as_uint(t)); }
    PT_DEV void pop(int sp, int &ref, float &t) {
        const uint2 e = base[sp * 64];
        ref = (int)e.x;
        t = __uint_as_float(e.y);
    }
};

// Closest hit inside one mesh's BLAS, local space (bvh_trace_local, intersection.cuh:344-435).
// `ref` is the root reference (>=0 inner node, <0 ~leaf); the root box was already
// tested by the caller.  Updates (tbest,u,v,slot) with strict `<`.
template <bool UNIFORM_LEAF>
PT_DEV void blas_closest(const KParams &K, int root_ref, bool alive, const RayO &r, LdsStack stk, float &tbest,
                         float &ub, float &vb, int &slotb) {
    if (UNIFORM_LEAF) {
        // root is a leaf and the reference to it is wave-uniform: triangle packets come
        // through scalar loads, every lane tests the same triangle
        const int leaf = __builtin_amdgcn_readfirstlane(~root_ref);
        const int2 lf = K.leaves[leaf];
        for (int i = 0; i < lf.y; ++i) {
            const int slot = lf.x + i;
            const float4 p0 = K.tris[slot * 3 + 0], p1 = K.tris[slot * 3 + 1], p2 = K.tris[slot * 3 + 2];
            float t, u, v;
            const bool ok = tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, tbest, t, u, v);
            if (alive && ok) {
                tbest = t;
                ub = u;
                vb = v;
                slotb = slot;
            }
        }
        return;
    }
    // "while-while" traversal: all lanes first descend through inner nodes until each holds a
    // leaf (or has no work left), THEN the wave tests leaves together, so the two code paths are
    // not serialised against each other every step (the per-lane order of visits is unchanged)
    int cur = root_ref;
    int sp = 0;
    bool active = alive;
    // the next subtree that can still contain a closer hit (E1)
    auto pop = [&]() {
        active = false;
        while (sp > 0) {
            --sp;
            int ref;
            float tE;
            stk.pop(sp, ref, tE);
            if (tE < tbest) {
                cur = ref;
                active = true;
                break;
            }
        }
    };
    while (active) {
        while (active && cur >= 0) {
            const float4 n0 = K.nodes[cur * 4 + 0], n1 = K.nodes[cur * 4 + 1], n2 = K.nodes[cur * 4 + 2],
                         n3 = K.nodes[cur * 4 + 3];
            float tL, tR;
            const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), r, tbest, tL);
            const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), r, tbest, tR);
            const int L = __float_as_int(n3.x), R = __float_as_int(n3.y);
            if (hL || hR) {
                const bool nearL = hL && (!hR || tL <= tR);
                if (nearL ? hR : hL) {
                    stk.push(sp, nearL ? R : L, nearL ? tR : tL);
                    ++sp;
                }
                cur = nearL ? L : R;
            } else {
                pop();
            }
        }
        if (active) { // cur is a leaf
            const int2 lf = K.leaves[~cur];
            // software-pipelined: packet i+1 is in flight while packet i is tested
            const float4 *tp = K.tris + (size_t)lf.x * 3;
            float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
            if (lf.y > 0) { // (an absent child's placeholder leaf is empty and owns no packet)
                p0 = tp[0];
                p1 = tp[1];
                p2 = tp[2];
            }
            for (int i = 0; i < lf.y; ++i) {
                const int nx = (i + 1 < lf.y) ? (i + 1) : i;
                const float4 q0 = tp[nx * 3 + 0], q1 = tp[nx * 3 + 1], q2 = tp[nx * 3 + 2];
                float t, u, v;
                if (tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, tbest, t, u, v)) {
                    tbest = t;
                    ub = u;
                    vb = v;
                    slotb = lf.x + i;
                }
                p0 = q0;
                p1 = q1;
                p2 = q2;
            }
            pop();
        }
    }
}

// Any hit inside one mesh's BLAS (bvh_any_hit_local, intersection.cuh:300-341), E4.
template <bool UNIFORM_LEAF>
PT_DEV bool blas_any(const KParams &K, int root_ref, bool alive, const RayO &r, float tMax, LdsStack stk) {
    bool found = false;
    if (UNIFORM_LEAF) {
        const int leaf = __builtin_amdgcn_readfirstlane(~root_ref);
        const int2 lf = K.leaves[leaf];
        for (int i = 0; i < lf.y; ++i) {
            const int slot = lf.x + i;
            const float4 p0 = K.tris[slot * 3 + 0], p1 = K.tris[slot * 3 + 1], p2 = K.tris[slot * 3 + 2];
            float t, u, v;
            found |= tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, tMax, t, u, v);
        }
        return alive && found;
    }
    int cur = root_ref;
    int sp = 0;
    bool active = alive;
    while (active) {
        if (cur >= 0) {
            const float4 n0 = K.nodes[cur * 4 + 0], n1 = K.nodes[cur * 4 + 1], n2 = K.nodes[cur * 4 + 2],
                         n3 = K.nodes[cur * 4 + 3];
            float tL, tR;
            const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), r, tMax, tL);
            const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), r, tMax, tR);
            const int L = __float_as_int(n3.x), R = __float_as_int(n3.y);
            if (hL && hR) {
                stk.push(sp, R, 0.0f);
                ++sp;
                cur = L;
                continue;
            }
            if (hL || hR) {
                cur = hL ? L : R;
                continue;
            }
        } else {
            const int2 lf = K.leaves[~cur];
            for (int i = 0; i < lf.y; ++i) {
                const int slot = lf.x + i;
                const float4 p0 = K.tris[slot * 3 + 0], p1 = K.tris[slot * 3 + 1], p2 = K.tris[slot * 3 + 2];
                float t, u, v;
                found |= tri_test(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, tMax, t, u, v);
            }
            if (found)
                break;
        }
        active = false;
        if (sp > 0) {
            --sp;
            float tE;
            stk.pop(sp, cur, tE);
            active = true;
        }
    }
    return found;
}

struct MeshHead {
    f3 bmin, bmax;
    int root_ref, flags; // flags bit0 has_transform, bit1 skipped by shadow rays (transmission > 0.5)
    int mesh;            // mesh id
};
PT_DEV MeshHead load_mesh_head(const KParams &K, int m) {
    const float4 a = K.mesh_recs[m * MESH_REC_F4 + 0], b = K.mesh_recs[m * MESH_REC_F4 + 1];
    MeshHead h;
    h.bmin = mk3(a.x, a.y, a.z);
    h.root_ref = __float_as_int(a.w);
    h.mesh = m;
    h.bmax = mk3(b.x, b.y, b.z);
    h.flags = __float_as_int(b.w);
    return h;
}
// local-space ray of a transformed instance (transformRayToLocal, intersection.cuh:284-297)
PT_DEV RayO local_ray_rows(const float4 i0, const float4 i1, const float4 i2, const RayO &w, float &dirScale) {
    const f3 lo = xform_point(i0, i1, i2, w.o);
    const f3 ld = xform_dir(i0, i1, i2, w.d);
    dirScale = length(ld);
    return make_ray(lo, normalize(ld));
}
PT_DEV RayO local_ray(const KParams &K, int m, const RayO &w, float &dirScale) {
    const float4 i0 = K.mesh_recs[m * MESH_REC_F4 + 2], i1 = K.mesh_recs[m * MESH_REC_F4 + 3],
                 i2 = K.mesh_recs[m * MESH_REC_F4 + 4];
    const f3 lo = xform_point(i0, i1, i2, w.o);
    const f3 ld = xform_dir(i0, i1, i2, w.d);
    dirScale = length(ld);
    return make_ray(lo, normalize(ld));
}

// One mesh of a TLAS leaf, closest hit (bvh_trace + the merge in traceRay,
// intersection.cuh:454-479,556-565).  GEOM 0: every BLAS is a single leaf.
template <int GEOM, bool UNIFORM>
PT_DEV void mesh_closest(const KParams &K, int m, bool alive, const RayO &w, LdsStack stk, Hit &best) {
    if (UNIFORM)
        m = __builtin_amdgcn_readfirstlane(m);
    const MeshHead mh = load_mesh_head(K, m);
    if (!(mh.flags & 1)) {
        float tE;
        const bool hb = alive && slab(mh.bmin, mh.bmax, w, T_FAR, tE);
        if (UNIFORM && !__builtin_amdgcn_ballot_w64(hb))
            return;
        if (GEOM == 0 || (UNIFORM && mh.root_ref < 0)) {
            // E3: a single-leaf BLAS has no inner boxes, so its triangles can run
            // straight on the global best
            float tb = best.t, ub = best.u, vb = best.v;
            int sb = -1;
            blas_closest<UNIFORM>(K, mh.root_ref, hb, w, stk, tb, ub, vb, sb);
            if (sb >= 0) {
                best.t = tb;
                best.t_local = tb;
                best.u = ub;
                best.v = vb;
                best.slot = sb;
                best.mesh = m;
            }
            return;
        }
        // inner boxes must be culled against THIS mesh's best only (the reference
        // restarts at 1e30 per mesh and its slab test is not conservative w.r.t. the
        // triangle test), then merged with strict `<`
        float tb = T_FAR, ub = 0.0f, vb = 0.0f;
        int sb = -1;
        blas_closest<false>(K, mh.root_ref, hb, w, stk, tb, ub, vb, sb);
        if (sb >= 0 && tb < best.t) {
            best.t = tb;
            best.t_local = tb;
            best.u = ub;
            best.v = vb;
            best.slot = sb;
            best.mesh = m;
        }
        return;
    }
    float dirScale;
    const RayO lr = local_ray(K, m, w, dirScale);
    float tE;
    const bool hb = alive && slab(mh.bmin, mh.bmax, lr, T_FAR, tE);
    if (UNIFORM && !__builtin_amdgcn_ballot_w64(hb))
        return;
    float tb = T_FAR, ub = 0.0f, vb = 0.0f;
    int sb = -1;
    if (GEOM == 0 || (UNIFORM && mh.root_ref < 0))
        blas_closest<UNIFORM>(K, mh.root_ref, hb, lr, stk, tb, ub, vb, sb);
    else
        blas_closest<false>(K, mh.root_ref, hb, lr, stk, tb, ub, vb, sb);
    if (sb >= 0) {
        const float tw = tb / dirScale;
        if (tw < best.t) {
            best.t = tw;
            best.t_local = tb;
            best.u = ub;
            best.v = vb;
            best.slot = sb;
            best.mesh = m;
        }
    }
}

template <int GEOM, bool UNIFORM>
PT_DEV bool mesh_any(const KParams &K, int m, bool alive, const RayO &w, float tMax, LdsStack stk) {
    if (UNIFORM)
        m = __builtin_amdgcn_readfirstlane(m);
    const MeshHead mh = load_mesh_head(K, m);
    if (mh.flags & 2) // transmission > 0.5: invisible to shadow rays (intersection.cuh:509-511)
        return false;
    RayO r = w;
    float tm = tMax;
    if (mh.flags & 1) {
        float dirScale;
        r = local_ray(K, m, w, dirScale);
        tm = tMax * dirScale;
    }
    float tE;
    const bool hb = alive && slab(mh.bmin, mh.bmax, r, tm, tE);
    if (UNIFORM && !__builtin_amdgcn_ballot_w64(hb))
        return false;
    if (GEOM == 0 || (UNIFORM && mh.root_ref < 0))
        return blas_any<UNIFORM>(K, mh.root_ref, hb, r, tm, stk);
    return blas_any<false>(K, mh.root_ref, hb, r, tm, stk);
}

// traceRay (intersection.cuh:526-605).  GEOM 0/1: the TLAS is a single leaf, so the
// mesh loop is wave-uniform.  GEOM 2: general TLAS with a small private stack.
template <int GEOM> PT_DEV Hit closest_hit(const KParams &K, bool alive, f3 o, f3 d, LdsStack stk) {
    Hit best;
    best.t = T_FAR;
    best.t_local = T_FAR;
    best.u = best.v = 0.0f;
    best.mesh = -1;
    best.slot = -1;
    const RayO w = make_ray(o, d);
    float tE;
    alive = alive && slab(tlas_bmin(K),
                          tlas_bmax(K), w, T_FAR, tE);
    if (GEOM < 2) {
        const int2 lf = K.tlas_leaves[~K.tlas_root_ref];
        for (int i = 0; i < lf.y; ++i)
            mesh_closest<GEOM, true>(K, K.tlas_mesh_ids[lf.x + i], alive, w, stk, best);
        return best;
    }
    int tstack_ref[24];
    float tstack_t[24];
    int sp = 0;
    int cur = K.tlas_root_ref;
    bool active = alive;
    while (active) {
        if (cur >= 0) {
            const float4 n0 = K.tlas_nodes[cur * 4 + 0], n1 = K.tlas_nodes[cur * 4 + 1], n2 = K.tlas_nodes[cur * 4 + 2],
                         n3 = K.tlas_nodes[cur * 4 + 3];
            float tL, tR;
            const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), w, best.t, tL);
            const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), w, best.t, tR);
            const int L = __float_as_int(n3.x), R = __float_as_int(n3.y);
            if (hL || hR) {
                const bool nearL = hL && (!hR || tL <= tR);
                if (nearL ? hR : hL) {
                    tstack_ref[sp] = nearL ? R : L;
                    tstack_t[sp] = nearL ? tR : tL;
                    ++sp;
                }
                cur = nearL ? L : R;
                continue;
            }
        } else {
            const int2 lf = K.tlas_leaves[~cur];
            for (int i = 0; i < lf.y; ++i)
                mesh_closest<GEOM, false>(K, K.tlas_mesh_ids[lf.x + i], true, w, stk, best);
        }
        active = false;
        while (sp > 0) {
            --sp;
            if (tstack_t[sp] < best.t) {
                cur = tstack_ref[sp];
                active = true;
                break;
            }
        }
    }
    return best;
}

// bvh_any_hit_tlas (intersection.cuh:481-524)
template <int GEOM> PT_DEV bool any_hit(const KParams &K, bool alive, f3 o, f3 d, float tMax, LdsStack stk) {
    const RayO w = make_ray(o, d);
    float tE;
    alive = alive && slab(tlas_bmin(K),
                          tlas_bmax(K), w, tMax, tE);
    bool found = false;
    if (GEOM < 2) {
        const int2 lf = K.tlas_leaves[~K.tlas_root_ref];
        for (int i = 0; i < lf.y; ++i) {
            found |= mesh_any<GEOM, true>(K, K.tlas_mesh_ids[lf.x + i], alive && !found, w, tMax, stk);
            if (!__builtin_amdgcn_ballot_w64(alive && !found))
                break;
        }
        return found;
    }
    int tstack_ref[24];
    int sp = 0;
    int cur = K.tlas_root_ref;
    bool active = alive;
    while (active) {
        if (cur >= 0) {
            const float4 n0 = K.tlas_nodes[cur * 4 + 0], n1 = K.tlas_nodes[cur * 4 + 1], n2 = K.tlas_nodes[cur * 4 + 2],
                         n3 = K.tlas_nodes[cur * 4 + 3];
            float tL, tR;
            const bool hL = slab(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), w, tMax, tL);
            const bool hR = slab(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), w, tMax, tR);
            const int L = __float_as_int(n3.x), R = __float_as_int(n3.y);
            if (hL && hR) {
                tstack_ref[sp++] = R;
                cur = L;
                continue;
            }
            if (hL || hR) {
                cur = hL ? L : R;
                continue;
            }
        } else {
            const int2 lf = K.tlas_leaves[~cur];
            for (int i = 0; i < lf.y && !found; ++i)
                found |= mesh_any<GEOM, false>(K, K.tlas_mesh_ids[lf.x + i], true, w, tMax, stk);
            if (found)
                break;
        }
        active = false;
        if (sp > 0) {
            cur = tstack_ref[--sp];
            active = true;
        }
    }
    return found;
}

// HitInfo fields derived from the winning triangle (intersection.cuh:380-392,465-476)
// The geometric normal of a packet, normalize(cross(e1, e2)) (intersection.cuh:386, 470), is a property of the triangle:
// computed once per (re)packing by the one function below and kept in the packets' three spare w words, instead of a cross
// product, a square root and three divisions (~100 VALU) in every shading iteration.
PT_DEV f3 packet_normal(float4 p1, float4 p2) { return normalize(cross(mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z))); }
__global__ void tri_normals_kernel(float4 *__restrict__ tris, int n_slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots)
        return;
    const f3 gn = packet_normal(tris[s * 3 + 1], tris[s * 3 + 2]);
    tris[s * 3 + 0].w = gn.x;
    tris[s * 3 + 1].w = gn.y;
    tris[s * 3 + 2].w = gn.z;
}

// Two-level node records: record i = {node i, node of its left child, node of its right child} (an absent or leaf child's
// part is zero), so that ONE fetch serves two steps of the binary walk -- the node's own child boxes, and then the boxes of
// whichever child the walk enters next (see descend2 in pt_render.hip.h).  Derived from the canonical child-pair nodes, which
// the refit / rebuild kernels keep writing; 192 B per inner node.
__global__ void expand_nodes_kernel(const float4 *__restrict__ nodes, float4 *__restrict__ nodes2, int n_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes * 3)
        return;
    const int n = i / 3, part = i % 3;
    int src = n;
    if (part) {
        const float4 refs = nodes[n * 4 + 3];
        src = __float_as_int(part == 1 ? refs.x : refs.y);
    }
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        nodes2[n * NODE2_F4 + part * 4 + k] = src >= 0 ? nodes[src * 4 + k] : z;
}

// HitInfo of a hit (intersection.cuh:382-396, 466-478) from the triangle's geometric normal and its mesh's flags
PT_DEV Surface make_surface_of(const KParams &K, const Hit &h, f3 gn, int flags, f3 o, f3 d, f3 *local_point) {
    Surface s;
    s.t = h.t;
    if (!(flags & 1)) {
        s.point = o + h.t * d;
        s.front_face = dot(d, gn) < 0.0f;
        s.normal = s.front_face ? gn : -gn;
        if (local_point)
            *local_point = s.point;
        return s;
    }
    const float4 *rec = K.mesh_recs + h.mesh * MESH_REC_F4;
    const f3 lo = xform_point(rec[2], rec[3], rec[4], o);
    const f3 ld = normalize(xform_dir(rec[2], rec[3], rec[4], d));
    const f3 lp = lo + h.t_local * ld;
    const bool lfront = dot(ld, gn) < 0.0f;
    const f3 ln = lfront ? gn : -gn;
    s.point = xform_point(rec[5], rec[6], rec[7], lp);
    const f3 wn = normalize(xform_dir(rec[8], rec[9], rec[10], ln));
    s.front_face = dot(d, wn) < 0.0f;
    s.normal = s.front_face ? wn : -wn;
    if (local_point)
        *local_point = lp;
    return s;
}
PT_DEV Surface make_surface(const KParams &K, const Hit &h, f3 o, f3 d, f3 *local_point, int *face_index) {
    const float *w = (const float *)(K.tris + h.slot * 3);
    const f3 gn = mk3(w[3], w[7], w[11]);
    const int flags = __float_as_int(K.mesh_recs[h.mesh * MESH_REC_F4 + 1].w);
    if (face_index)
        *face_index = K.slot_face[h.slot].w;
    return make_surface_of(K, h, gn, flags, o, d, local_point);
}

struct LightRec {
    f3 position, direction, color;
    int type;
    float intensity, range, inner, outer, radius;
};
template <bool LDS = false> PT_DEV LightRec load_light(const float4 *__restrict__ L, int i) {
    const float4 a = ld4<LDS>(L, i * 4 + 0), b = ld4<LDS>(L, i * 4 + 1), c = ld4<LDS>(L, i * 4 + 2), d = ld4<LDS>(L, i * 4 + 3);
    LightRec l;
    l.position = mk3(a.x, a.y, a.z);
    l.type = __float_as_int(a.w);
    l.direction = mk3(b.x, b.y, b.z);
    l.intensity = b.w;
    l.color = mk3(c.x, c.y, c.z);
    l.range = c.w;
    l.inner = d.x;
    l.outer = d.y;
    l.radius = d.z;
    return l;
}

// ---------------------------------------------------------------------------------
// the render loop itself: pt_render.hip.h
// ---------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------
// XORWOW initialisation: state(seed) advanced by `global pixel index` subsequences of
// 2^67 draws.  jump[k] = (step^(2^67))^(2^k) as 160 columns x 5 words; a lane applies
// the matrices selected by the bits of its pixel index.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xorwow_init_kernel(uint32_t *rng, int width, int rows, int y0, int il_period,
                                                          int il_phase, uint32_t d0, uint32_t s0, uint32_t s1, uint32_t s2,
                                                          uint32_t s3, uint32_t s4, const uint32_t *__restrict__ jump,
                                                          int n_jump) {
    const size_t npix = (size_t)rows * width;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix)
        return;
    const int yl = (int)(i / (size_t)width), x = (int)(i % (size_t)width);
    unsigned long long n = (unsigned long long)global_row(yl, y0, il_period, il_phase) * (unsigned long long)width + x; // global y*W+x
    uint32_t v[5] = {s0, s1, s2, s3, s4};
    for (int k = 0; k < n_jump && n; ++k, n >>= 1) {
        if (!(n & 1ull))
            continue;
        const uint32_t *M = jump + (size_t)k * 800;
        uint32_t a[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < 5; ++w) {
            uint32_t bits = v[w];
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t *c = M + (w * 32 + b) * 5;
                a[0] ^= c[0];
                a[1] ^= c[1];
                a[2] ^= c[2];
                a[3] ^= c[3];
                a[4] ^= c[4];
            }
        }
        for (int w = 0; w < 5; ++w)
            v[w] = a[w];
    }
    rng[i] = d0;
    rng[npix + i] = v[0];
    rng[2 * npix + i] = v[1];
    rng[3 * npix + i] = v[2];
    rng[4 * npix + i] = v[3];
    rng[5 * npix + i] = v[4];
}

// PMODE 3: the heads of the mesh records in TLAS-leaf order, refreshed before a frame (root boxes move under a GPU
// refit, the shadow-skip flag with the materials): a lane's leaf is then ONE level of loads away instead of two
// Entry j: [0] {box of the FIRST pass, root}, [1] {.., flags}, [2..4] inverse rows, [5], [6] an instance's own head.
// The first-pass box of an untransformed mesh is its root box (the test is the reference's); that of an instance is a
// world-space box that provably contains every ray its local-space test can accept (`pre`, host-computed; DESIGN.md
// 3.10) -- or everything, while the device-side boxes have moved since the host computed it (pre_ok == 0).
__global__ void gather_tlas_heads_kernel(const float4 *__restrict__ mesh_recs, const float4 *__restrict__ pre,
                                         const int *__restrict__ ids, int n, float4 *__restrict__ heads, int pre_ok) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const int m = ids[j];
    float4 h0 = mesh_recs[m * MESH_REC_F4 + 0], h1 = mesh_recs[m * MESH_REC_F4 + 1];
    heads[TLAS_HEAD_F4 * j + 5] = h0;
    heads[TLAS_HEAD_F4 * j + 6] = h1;
    if (__float_as_int(h1.w) & 1) {
        const float4 p0 = pre[2 * m], p1 = pre[2 * m + 1];
        const float big = 3.0e38f;
        h0 = make_float4(pre_ok ? p0.x : -big, pre_ok ? p0.y : -big, pre_ok ? p0.z : -big, h0.w);
        h1 = make_float4(pre_ok ? p1.x : big, pre_ok ? p1.y : big, pre_ok ? p1.z : big, h1.w);
    }
    heads[TLAS_HEAD_F4 * j + 0] = h0;
    heads[TLAS_HEAD_F4 * j + 1] = h1;
    for (int k = 2; k < 5; ++k)
        heads[TLAS_HEAD_F4 * j + k] = mesh_recs[m * MESH_REC_F4 + k];
}

// canonical {d,v0..v4}-per-pixel order <-> the private planar layout
__global__ void rng_planar_to_aos(const uint32_t *planar, uint32_t *aos, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix)
        return;
    for (int k = 0; k < 6; ++k)
        aos[i * 6 + k] = planar[(size_t)k * npix + i];
}
__global__ void rng_aos_to_planar(const uint32_t *aos, uint32_t *planar, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix)
        return;
    for (int k = 0; k < 6; ++k)
        planar[(size_t)k * npix + i] = aos[i * 6 + k];
}

// ---------------------------------------------------------------------------------
// trace_single_ray_kernel, batched: one lane per ray
// ---------------------------------------------------------------------------------
struct HitOut { // == ptrt_hit (include/ptrt.h)
    int hit;
    float t;
    float point[3], normal[3];
    int mesh_index, front_face;
    float u, v;
    int face_index;
    float local_point[3];
};
template <int GEOM>
__global__ __launch_bounds__(64) void trace_rays_kernel(const KParams K, const float *origins, const float *dirs, int n,
                                                        HitOut *out) {
    extern __shared__ uint2 lds_stack[];
    LdsStack stk{lds_stack + threadIdx.x};
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < n;
    const int j = live ? i : 0;
    const f3 o = mk3(origins[j * 3], origins[j * 3 + 1], origins[j * 3 + 2]);
    const f3 d = mk3(dirs[j * 3], dirs[j * 3 + 1], dirs[j * 3 + 2]);
    const Hit h = closest_hit<GEOM>(K, live, o, d, stk);
    if (!live)
        return;
    HitOut r;
    if (h.mesh < 0) { // HitInfo() defaults (intersection.cuh:122-124)
        r.hit = 0;
        r.t = 1e30f;
        r.point[0] = r.point[1] = r.point[2] = 0.0f;
        r.normal[0] = r.normal[1] = r.normal[2] = 0.0f;
        r.mesh_index = -1;
        r.front_face = 1;
        r.u = r.v = 0.0f;
        r.face_index = -1;
        r.local_point[0] = r.local_point[1] = r.local_point[2] = 0.0f;
    } else {
        f3 lp;
        int face;
        const Surface s = make_surface(K, h, o, d, &lp, &face);
        r.hit = 1;
        r.t = h.t;
        r.point[0] = s.point.x; r.point[1] = s.point.y; r.point[2] = s.point.z;
        r.normal[0] = s.normal.x; r.normal[1] = s.normal.y; r.normal[2] = s.normal.z;
        r.mesh_index = h.mesh;
        r.front_face = s.front_face ? 1 : 0;
        r.u = h.u;
        r.v = h.v;
        r.face_index = face;
        r.local_point[0] = lp.x; r.local_point[1] = lp.y; r.local_point[2] = lp.z;
    }
    out[i] = r;
}

// exhaustive check of rcp_ieee: every fp32 bit pattern, against the compiler's IEEE division.
// out[0] = number of mismatching inputs, out[1..8] = first few offending bit patterns.
__global__ void rcp_check_kernel(unsigned int *out) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float y = __uint_as_float((uint32_t)b);
        const float want = 1.0f / y;
        const float got = rcp_ieee(y);
        const uint32_t wu = __float_as_uint(want), gu = __float_as_uint(got);
        const bool both_nan = (want != want) && (got != got);
        if (wu != gu && !both_nan) {
            const unsigned int k = atomicAdd(&out[0], 1u);
            if (k < 8)
                out[1 + k] = (uint32_t)b;
        }
    }
}

// exhaustive check of sqrt_ieee: every fp32 bit pattern against the compiler's IEEE sqrtf.  out[0] = mismatches of sqrt_ieee,
// out[1] = mismatches of the unguarded core over the guarded range, out[2..8] = first offending inputs of sqrt_ieee.
__global__ void sqrt_check_kernel(unsigned int *out) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = __builtin_sqrtf(x);
        const float got = sqrt_ieee(x);
        const uint32_t wu = __float_as_uint(want), gu = __float_as_uint(got);
        if (wu != gu && !((want != want) && (got != got))) {
            const unsigned int k = atomicAdd(&out[0], 1u);
            if (k < 7)
                out[2 + k] = (uint32_t)b;
        }
        const bool mid = ((uint32_t)b - 0x21800000u) < (0x5d800000u - 0x21800000u);
        if (mid && __float_as_uint(sqrt_core(x)) != wu)
            atomicAdd(&out[1], 1u);
    }
}

// exhaustive check of div3's core: for the divisors with significand index [first, first + count) (t = 1.m) EVERY
// numerator significand (a = 1.m', 2^23 of them), against the compiler's IEEE division; plus, per divisor, the same numerators
// through div3 itself with exponents chosen by `mode` (0: a in [1, 2); 1: a = m' * 2^-149 .. subnormal and tiny, t scaled by
// 2^-30; 2: a scaled by 2^70, t by 2^-35: the fallback's ranges; 3: as 0 with the RAW v_rcp_f32 instead of the refined reciprocal -- the negative control).
// out[0] = mismatches, out[1..8] = first offending
// {a bits, t bits} pairs (4 of them).
__global__ void div3_check_kernel(uint32_t first, uint32_t count, int mode, unsigned int *out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count)
        return;
    float t = __uint_as_float(0x3f800000u | ((first + j) & 0x7fffffu));
    if (mode == 1)
        t *= 0x1p-30f;
    if (mode == 2)
        t *= 0x1p-35f;
    const float r0 = __builtin_amdgcn_rcpf(t);
    const float r = fma_(fma_(-t, r0, 1.0f), r0, r0);
    unsigned int bad = 0;
    for (uint32_t m = 0; m < (1u << 23); m += 3) {
        f3 a;
        if (mode == 0 || mode == 3)
            a = mk3(__uint_as_float(0x3f800000u | m), __uint_as_float(0x3f800000u | ((m + 1) & 0x7fffffu)),
                    __uint_as_float(0x3f800000u | ((m + 2) & 0x7fffffu)));
        else if (mode == 1)
            a = mk3(__uint_as_float(m), -__uint_as_float(0x00800000u + m * 16u), __uint_as_float(0x0c000000u + m));
        else
            a = mk3(__uint_as_float(0x3f800000u | m) * 0x1p70f, __uint_as_float(0x7f000000u + 2u * m), -__uint_as_float(0x3f800000u | m));
        const f3 want = f3{a.x / t, a.y / t, a.z / t};
        const f3 got = mode == 0 ? f3{div3_core(a.x, t, r), div3_core(a.y, t, r), div3_core(a.z, t, r)}
                       : mode == 3 ? f3{div3_core(a.x, t, r0), div3_core(a.y, t, r0), div3_core(a.z, t, r0)} : div3(a, t);
        const bool ne = (__float_as_uint(want.x) != __float_as_uint(got.x) && !(want.x != want.x && got.x != got.x)) ||
                        (__float_as_uint(want.y) != __float_as_uint(got.y) && !(want.y != want.y && got.y != got.y)) ||
                        (__float_as_uint(want.z) != __float_as_uint(got.z) && !(want.z != want.z && got.z != got.z));
        if (ne && bad < 4) {
            const unsigned int k = atomicAdd(&out[0], 1u);
            if (k < 4) {
                out[1 + 2 * k] = __float_as_uint(a.x);
                out[2 + 2 * k] = __float_as_uint(t);
            }
            ++bad;
        }
    }
}

// deterministic-math probe for tests: op 0 sin, 1 cos, 2 exp, 3 log, 4 pow
__global__ void detmath_kernel(int op, const float *x, const float *y, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float r;
    switch (op) {
    case 0: r = det_sin(x[i]); break;
    case 1: r = det_cos(x[i]); break;
    case 2: r = det_exp(x[i]); break;
    case 3: r = det_log(x[i]); break;
    case 5: r = det_atan2(x[i], y[i]); break;
    case 6: r = det_acos(x[i]); break;
    default: r = det_pow(x[i], y[i]); break;
    }
    out[i] = r;
}

} // namespace pt
