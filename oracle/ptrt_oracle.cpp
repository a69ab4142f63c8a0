/*
 * oracle/ptrt_oracle.cpp -- TEST INFRASTRUCTURE.  CPU restatement of the reference's
 * path-tracing render loop.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (ptrt-game-engine_amd/) never does.
 *
 * PARITY UNPINNED except for what oracle/_ref pins: the reference
 * (Mark-Rindler/PTRT-game-engine) ships no tests, golden images or known-answer
 * vectors for this path.  Its curand-free headers DO compile in this image
 * against the real CUDA runtime headers it ships (oracle/ref_probe.cpp ->
 * oracle/_ref/ref_probe: blue-noise generator, TAA jitter table, Light), and
 * tests/test_ref_probe.py holds this file's getTAAJitter and the blue-noise
 * table the renders feed on to that build bit for bit.  Everything else needs
 * <curand_kernel.h> (absent; writing a stand-in is not allowed) and is a
 * restatement pinned only by (i) constants and tables that are literal in the
 * reference's source, (ii) the behavioural facts recorded in SURVEY.md
 * (Appendix B), and (iii) for the XORWOW recurrence and its 2^67 jump,
 * rocRAND's independent implementation of the same generator
 * (tests/golden/xorwow_rocrand_kat.json).  cuRAND's seed-scrambling constants
 * are restated from the published algorithm and are NOT verified against a
 * CUDA toolkit.
 *
 * What is restated (reference file:line at each function):
 *   path_trace_kernel   src/pathtracer/scene/scene_kernels.cuh:122-194
 *   tracePath & friends src/pathtracer/rendering/path_logic.cuh (live functions)
 *   traceRay, BVH       src/pathtracer/math/intersection.cuh
 *   sampling / pdf      src/pathtracer/math/{sampling,pdf}.cuh
 *   PBR helpers         src/pathtracer/rendering/{pbr_utils,render_utils}.cuh
 *   jitter              src/pathtracer/rendering/taa.cuh, math/sampling.cuh:15-43
 *   camera rays         src/pathtracer/scene/camera.cuh:23-30,156-205
 *   tonemap_kernel      src/pathtracer/scene/scene.cuh:2004-2047
 *   cuRAND XORWOW       third-party (CUDA Toolkit curand_kernel.h), see above
 *
 * Arithmetic contract (shared, by construction, with the HIP kernels so that the
 * comparison is bit-exact, not statistical):
 *   - built with -ffp-contract=off: every a*b+c in the source below is a
 *     rounded multiply followed by a rounded add, EXCEPT
 *   - dot(), cross() and length_squared() use the fused form nvcc's default
 *     -fmad=true produces for them:  dot = fma(az,bz, fma(ay,by, ax*bx)),
 *     cross.x = fma(ay,bz, -(az*by)) etc.; and
 *   - sinf/cosf/expf/logf/powf are the deterministic versions in detmath.h,
 *     rsqrtf(x) is 1/sqrtf(x), __frcp_rn(x) is 1/x, powf(x,2) is x*x.
 */
#include "../include/ptrt.h"
#include "detmath.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// vec3 (src/common/vec3.cuh)
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    V3(const ptrt_vec3 &p) : x(p.x), y(p.y), z(p.z) {}
};
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3 &a, const V3 &b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(const V3 &a, float t) { return V3(a.x * t, a.y * t, a.z * t); }
inline V3 operator*(float t, const V3 &a) { return a * t; } // vec3.cuh:126-128
inline V3 operator/(const V3 &a, float t) { return V3(a.x / t, a.y / t, a.z / t); }
inline V3 operator/(const V3 &a, const V3 &b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 operator+(const V3 &a, float t) { return a + V3(t); } // implicit vec3(float)
inline V3 operator-(const V3 &a, float t) { return a - V3(t); }

// fused forms, see the arithmetic contract above (vec3.cuh:130-142)
inline float dot(const V3 &a, const V3 &b) { return dm_fma(a.z, b.z, dm_fma(a.y, b.y, a.x * b.x)); }
inline V3 cross(const V3 &a, const V3 &b) {
    return V3(dm_fma(a.y, b.z, -(a.z * b.y)), dm_fma(a.z, b.x, -(a.x * b.z)),
              dm_fma(a.x, b.y, -(a.y * b.x)));
}
inline float length_squared(const V3 &v) { return dot(v, v); }
inline float length(const V3 &v) { return sqrtf(dot(v, v)); }
inline V3 normalize(const V3 &v) { // vec3.cuh:110-113
    float len = length(v);
    return (len > 0) ? (v / len) : V3(0, 0, 0);
}
inline V3 lerp(const V3 &a, const V3 &b, float t) { return (1.0f - t) * a + t * b; } // vec3.cuh:156
inline float clamp01(float x) { return dm_min(dm_max(x, 0.0f), 1.0f); }              // render_utils.cuh:33
inline float clampf(float x, float lo, float hi) { return dm_min(dm_max(x, lo), hi); } // mathutils.cuh:35
inline V3 clampv(const V3 &v, float lo, float hi) {                                    // vec3.cuh:160
    return V3(dm_min(dm_max(v.x, lo), hi), dm_min(dm_max(v.y, lo), hi), dm_min(dm_max(v.z, lo), hi));
}
inline V3 reflectVec(const V3 &I, const V3 &N) { return I - 2.0f * dot(I, N) * N; } // render_utils.cuh:41

constexpr float PI_F = 3.14159265358979323846f;     // mathutils.cuh:13
constexpr float TWO_PI_F = 6.28318530717958647692f; // mathutils.cuh:14
constexpr float EPSILON_F = 1e-6f;                  // mathutils.cuh:19

// ---------------------------------------------------------------------------
// cuRAND XORWOW (third-party; published algorithm restated)
// ---------------------------------------------------------------------------
struct Xorwow {
    uint32_t d;
    uint32_t v[5];
};

inline uint32_t xorwow_next(Xorwow &s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1];
    s.v[1] = s.v[2];
    s.v[2] = s.v[3];
    s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}
// curand_uniform: (0,1]
inline float xorwow_uniform(Xorwow &s) {
    return (float)xorwow_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// 160x160 matrix over GF(2), stored as the images of the 160 basis vectors.
struct GF2Mat {
    uint32_t col[160][5];
};
inline void gf2_apply(const GF2Mat &m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t acc[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w) {
        uint32_t bits = in[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t *c = m.col[w * 32 + b];
            for (int k = 0; k < 5; ++k)
                acc[k] ^= c[k];
        }
    }
    for (int k = 0; k < 5; ++k)
        out[k] = acc[k];
}
inline void gf2_mul(const GF2Mat &a, const GF2Mat &b, GF2Mat &out) { // out = a o b
    GF2Mat tmp;
    for (int j = 0; j < 160; ++j)
        gf2_apply(a, b.col[j], tmp.col[j]);
    out = tmp;
}
void gf2_step_matrix(GF2Mat &m) { // one xorwow_next on the v[] part
    for (int j = 0; j < 160; ++j) {
        Xorwow s;
        s.d = 0;
        for (int k = 0; k < 5; ++k)
            s.v[k] = 0;
        s.v[j / 32] = 1u << (j % 32);
        xorwow_next(s);
        for (int k = 0; k < 5; ++k)
            m.col[j][k] = s.v[k];
    }
}
// (step)^(2^67): one cuRAND "subsequence"
const GF2Mat &subsequence_matrix() {
    static GF2Mat m;
    static bool ready = false;
    if (!ready) {
        gf2_step_matrix(m);
        for (int i = 0; i < 67; ++i)
            gf2_mul(m, m, m);
        ready = true;
    }
    return m;
}

struct SeedConstants {
    uint32_t xor0, xor1, mul0, mul1;
};
// curand_init(seed, subsequence, 0): published cuRAND scrambling (UNVERIFIED against a toolkit).
constexpr SeedConstants CURAND_CONSTANTS = {0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u};

void xorwow_seed(Xorwow &s, unsigned long long seed, const SeedConstants &c) {
    uint32_t s0 = ((uint32_t)seed) ^ c.xor0;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ c.xor1;
    uint32_t t0 = c.mul0 * s0;
    uint32_t t1 = c.mul1 * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
}
// advance by `n` subsequences (d is unchanged: 362437 * 2^67 = 0 mod 2^32)
void xorwow_skip_subsequences(Xorwow &s, unsigned long long n) {
    GF2Mat p = subsequence_matrix();
    while (n) {
        if (n & 1ull) {
            uint32_t out[5];
            gf2_apply(p, s.v, out);
            for (int k = 0; k < 5; ++k)
                s.v[k] = out[k];
        }
        n >>= 1;
        if (n)
            gf2_mul(p, p, p);
    }
}

// ---------------------------------------------------------------------------
// scene access
// ---------------------------------------------------------------------------
struct Ray { // src/common/ray.cuh
    V3 orig, dir;
    bool spec;
    Ray() : spec(false) {}
    Ray(const V3 &o, const V3 &d) : orig(o), dir(d), spec(false) {}
    Ray(const V3 &o, const V3 &d, bool s) : orig(o), dir(d), spec(s) {}
};

struct RayOptimized { // intersection.cuh:39-88
    V3 origin, direction, invDirection;
    int dirSign[3];
    RayOptimized(const V3 &o, const V3 &d) {
        origin = o;
        direction = d;
        invDirection.x = (fabsf(d.x) > 1e-8f) ? (1.0f / d.x) : ((d.x >= 0) ? 1e30f : -1e30f);
        invDirection.y = (fabsf(d.y) > 1e-8f) ? (1.0f / d.y) : ((d.y >= 0) ? 1e30f : -1e30f);
        invDirection.z = (fabsf(d.z) > 1e-8f) ? (1.0f / d.z) : ((d.z >= 0) ? 1e30f : -1e30f);
        dirSign[0] = invDirection.x < 0 ? 1 : 0;
        dirSign[1] = invDirection.y < 0 ? 1 : 0;
        dirSign[2] = invDirection.z < 0 ? 1 : 0;
    }
    V3 at(float t) const { return origin + t * direction; }
};

struct HitInfo { // intersection.cuh:108-132
    bool hit = false;
    float t = 1e30f;
    V3 point, normal;
    int mesh_index = -1;
    bool front_face = true;
    float u = 0.0f, v = 0.0f;
    int face_index = -1;
    V3 localPoint;
    void set_face_normal(const V3 &rayDir, const V3 &outward) {
        front_face = dot(rayDir, outward) < 0.0f;
        normal = front_face ? outward : -outward;
    }
};

constexpr int BVH_STACK_SIZE = 24; // intersection.cuh:17

// intersection.cuh:136-172
inline bool aabb_hit_fast(const ptrt_bvh_node &n, const RayOptimized &ray, float tMax) {
    float t0x = (n.bmin.x - ray.origin.x) * ray.invDirection.x;
    float t1x = (n.bmax.x - ray.origin.x) * ray.invDirection.x;
    if (ray.dirSign[0]) { float tmp = t0x; t0x = t1x; t1x = tmp; }
    float t0y = (n.bmin.y - ray.origin.y) * ray.invDirection.y;
    float t1y = (n.bmax.y - ray.origin.y) * ray.invDirection.y;
    if (ray.dirSign[1]) { float tmp = t0y; t0y = t1y; t1y = tmp; }
    float tmin = dm_max(t0x, t0y);
    float tmax = dm_min(t1x, t1y);
    if (tmin > tmax)
        return false;
    float t0z = (n.bmin.z - ray.origin.z) * ray.invDirection.z;
    float t1z = (n.bmax.z - ray.origin.z) * ray.invDirection.z;
    if (ray.dirSign[2]) { float tmp = t0z; t0z = t1z; t1z = tmp; }
    tmin = dm_max(tmin, t0z);
    tmax = dm_min(tmax, t1z);
    return (tmax >= 0.0f) && (tmin <= tmax) && (tmin < tMax);
}
// intersection.cuh:175-216
inline bool aabb_hit_fast_t(const ptrt_bvh_node &n, const RayOptimized &ray, float tMax, float &tHit) {
    float t0x = (n.bmin.x - ray.origin.x) * ray.invDirection.x;
    float t1x = (n.bmax.x - ray.origin.x) * ray.invDirection.x;
    if (ray.dirSign[0]) { float tmp = t0x; t0x = t1x; t1x = tmp; }
    float t0y = (n.bmin.y - ray.origin.y) * ray.invDirection.y;
    float t1y = (n.bmax.y - ray.origin.y) * ray.invDirection.y;
    if (ray.dirSign[1]) { float tmp = t0y; t0y = t1y; t1y = tmp; }
    float tmin = dm_max(t0x, t0y);
    float tmax = dm_min(t1x, t1y);
    if (tmin > tmax)
        return false;
    float t0z = (n.bmin.z - ray.origin.z) * ray.invDirection.z;
    float t1z = (n.bmax.z - ray.origin.z) * ray.invDirection.z;
    if (ray.dirSign[2]) { float tmp = t0z; t0z = t1z; t1z = tmp; }
    tmin = dm_max(tmin, t0z);
    tmax = dm_min(tmax, t1z);
    if ((tmax < 0.0f) || (tmin > tmax) || (tmin >= tMax))
        return false;
    tHit = dm_max(tmin, 0.0f);
    return true;
}
// intersection.cuh:219-255
inline bool triangle_intersect_fast(const V3 &v0, const V3 &v1, const V3 &v2, const RayOptimized &ray,
                                    float tMax, float &t_out, float &u_out, float &v_out) {
    V3 e1 = v1 - v0;
    V3 e2 = v2 - v0;
    V3 h = cross(ray.direction, e2);
    float a = dot(e1, h);
    if (fabsf(a) < EPSILON_F)
        return false;
    float f = 1.0f / a;
    V3 s = ray.origin - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f)
        return false;
    V3 q = cross(s, e1);
    float v = f * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f)
        return false;
    float t = f * dot(e2, q);
    if (t > EPSILON_F && t < tMax) {
        t_out = t;
        u_out = u;
        v_out = v;
        return true;
    }
    return false;
}

// intersection.cuh:258-281 (row-major use of the 16 floats)
inline V3 transformPoint(const float *m, const V3 &p) {
    return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
inline V3 transformDirection(const float *m, const V3 &d) {
    return V3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
              m[8] * d.x + m[9] * d.y + m[10] * d.z);
}
inline V3 transformNormal(const float *m, const V3 &n) { return normalize(transformDirection(m, n)); }

struct Counters {
    uint64_t extension = 0, shadow = 0, paths = 0;
    // light samples whose value is exactly zero whatever their visibility (bookkeeping for the kernel's "counted, not
    // walked" shadow rays; the reference traces them like any other and nothing here depends on the count)
    uint64_t shadow_zero = 0;
};

// intersection.cuh:300-341
bool bvh_any_hit_local(const RayOptimized &localRay, const ptrt_mesh_desc &M, float tMax) {
    if (!M.nodes || M.node_count == 0)
        return false;
    int stack[BVH_STACK_SIZE];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const int ni = stack[--sp];
        const ptrt_bvh_node &N = M.nodes[ni];
        if (!aabb_hit_fast(N, localRay, tMax))
            continue;
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                int fidx = M.prim_indices[N.start + i];
                const ptrt_tri idx = M.faces[fidx];
                V3 v0 = M.verts[idx.v0], v1 = M.verts[idx.v1], v2 = M.verts[idx.v2];
                float t, u, v;
                if (triangle_intersect_fast(v0, v1, v2, localRay, tMax, t, u, v)) {
                    if (t > 1e-5f)
                        return true;
                }
            }
        } else {
            if (N.right >= 0 && sp < BVH_STACK_SIZE)
                stack[sp++] = N.right;
            if (N.left >= 0 && sp < BVH_STACK_SIZE)
                stack[sp++] = N.left;
        }
    }
    return false;
}

// intersection.cuh:344-435
HitInfo bvh_trace_local(const RayOptimized &localRay, const ptrt_mesh_desc &M) {
    HitInfo out;
    out.hit = false;
    out.t = 1e30f;
    if (!M.nodes || M.node_count == 0 || !M.prim_indices)
        return out;
    int stack[BVH_STACK_SIZE];
    int sp = 0;
    int ni = 0;
    while (true) {
        const ptrt_bvh_node &N = M.nodes[ni];
        if (!aabb_hit_fast(N, localRay, out.t)) {
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                const int fidx = M.prim_indices[N.start + i];
                const ptrt_tri tri = M.faces[fidx];
                V3 v0 = M.verts[tri.v0], v1 = M.verts[tri.v1], v2 = M.verts[tri.v2];
                float tHit, uHit, vHit;
                if (triangle_intersect_fast(v0, v1, v2, localRay, out.t, tHit, uHit, vHit)) {
                    if (tHit > 1e-5f) {
                        out.hit = true;
                        out.t = tHit;
                        out.localPoint = localRay.at(tHit);
                        out.point = out.localPoint;
                        V3 e1 = v1 - v0;
                        V3 e2 = v2 - v0;
                        V3 geom_normal = normalize(cross(e1, e2));
                        out.set_face_normal(localRay.direction, geom_normal);
                        out.u = uHit;
                        out.v = vHit;
                        out.face_index = fidx;
                    }
                }
            }
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        const int L = N.left, R = N.right;
        float tL = 0.f, tR = 0.f;
        bool hL = (L >= 0) && aabb_hit_fast_t(M.nodes[L], localRay, out.t, tL);
        bool hR = (R >= 0) && aabb_hit_fast_t(M.nodes[R], localRay, out.t, tR);
        if (!hL && !hR) {
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        int nearIdx, farIdx;
        bool hitFar;
        if (hL && (!hR || tL <= tR)) {
            nearIdx = L; farIdx = R; hitFar = hR;
        } else {
            nearIdx = R; farIdx = L; hitFar = hL;
        }
        if (hitFar && sp < BVH_STACK_SIZE)
            stack[sp++] = farIdx;
        ni = nearIdx;
    }
    return out;
}

inline RayOptimized transformRayToLocal(const RayOptimized &w, const float *inv) { // intersection.cuh:284-289
    V3 lo = transformPoint(inv, w.origin);
    V3 ld = transformDirection(inv, w.direction);
    return RayOptimized(lo, normalize(ld));
}
inline float getDirectionScale(const float *inv, const V3 &d) { return length(transformDirection(inv, d)); }

// intersection.cuh:438-450
bool bvh_any_hit(const RayOptimized &worldRay, const ptrt_mesh_desc &M, float worldTMax) {
    if (!M.has_transform)
        return bvh_any_hit_local(worldRay, M, worldTMax);
    RayOptimized localRay = transformRayToLocal(worldRay, M.inverse);
    float dirScale = getDirectionScale(M.inverse, worldRay.direction);
    float localTMax = worldTMax * dirScale;
    return bvh_any_hit_local(localRay, M, localTMax);
}
// intersection.cuh:454-479
HitInfo bvh_trace(const RayOptimized &worldRay, const ptrt_mesh_desc &M) {
    if (!M.has_transform)
        return bvh_trace_local(worldRay, M);
    RayOptimized localRay = transformRayToLocal(worldRay, M.inverse);
    HitInfo result = bvh_trace_local(localRay, M);
    if (result.hit) {
        result.point = transformPoint(M.world, result.localPoint);
        float dirScale = getDirectionScale(M.inverse, worldRay.direction);
        result.t = result.t / dirScale;
        V3 localNormal = result.normal;
        V3 worldNormal = transformNormal(M.normal, localNormal);
        result.front_face = dot(worldRay.direction, worldNormal) < 0.0f;
        result.normal = result.front_face ? worldNormal : -worldNormal;
    }
    return result;
}

// intersection.cuh:481-524
bool bvh_any_hit_tlas(const Ray &ray, float tMax, const ptrt_scene_desc &S, Counters &cnt) {
    cnt.shadow++;
    if (!S.tlas_nodes)
        return false;
    RayOptimized optRay(ray.orig, ray.dir);
    if (!aabb_hit_fast(S.tlas_nodes[0], optRay, tMax))
        return false;
    int stack[BVH_STACK_SIZE];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const int ni = stack[--sp];
        const ptrt_bvh_node &N = S.tlas_nodes[ni];
        if (!aabb_hit_fast(N, optRay, tMax))
            continue;
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                int mesh_id = S.tlas_mesh_indices[N.start + i];
                float trans = S.materials.transmission[mesh_id];
                if (trans > 0.5f)
                    continue;
                if (bvh_any_hit(optRay, S.meshes[mesh_id], tMax))
                    return true;
            }
        } else {
            if (N.right >= 0 && sp < BVH_STACK_SIZE)
                stack[sp++] = N.right;
            if (N.left >= 0 && sp < BVH_STACK_SIZE)
                stack[sp++] = N.left;
        }
    }
    return false;
}

// intersection.cuh:526-605
HitInfo traceRay(const Ray &ray, const ptrt_scene_desc &S, Counters &cnt) {
    cnt.extension++;
    HitInfo best;
    best.hit = false;
    best.t = 1e30f;
    if (!S.tlas_nodes)
        return best;
    RayOptimized optRay(ray.orig, ray.dir);
    if (!aabb_hit_fast(S.tlas_nodes[0], optRay, best.t))
        return best;
    int stack[BVH_STACK_SIZE];
    int sp = 0;
    int ni = 0;
    while (true) {
        const ptrt_bvh_node &N = S.tlas_nodes[ni];
        if (!aabb_hit_fast(N, optRay, best.t)) {
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                int mesh_id = S.tlas_mesh_indices[N.start + i];
                HitInfo h = bvh_trace(optRay, S.meshes[mesh_id]);
                if (h.hit && h.t < best.t) {
                    best = h;
                    best.mesh_index = mesh_id;
                }
            }
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        const int L = N.left, R = N.right;
        float tL = 0.f, tR = 0.f;
        bool hL = (L >= 0) && aabb_hit_fast_t(S.tlas_nodes[L], optRay, best.t, tL);
        bool hR = (R >= 0) && aabb_hit_fast_t(S.tlas_nodes[R], optRay, best.t, tR);
        if (!hL && !hR) {
            if (sp == 0)
                break;
            ni = stack[--sp];
            continue;
        }
        int nearIdx, farIdx;
        bool hitFar;
        if (hL && (!hR || tL <= tR)) {
            nearIdx = L; farIdx = R; hitFar = hR;
        } else {
            nearIdx = R; farIdx = L; hitFar = hL;
        }
        if (hitFar && sp < BVH_STACK_SIZE)
            stack[sp++] = farIdx;
        ni = nearIdx;
    }
    return best;
}

// ---------------------------------------------------------------------------
// sampling (math/sampling.cuh)
// ---------------------------------------------------------------------------
// sampling.cuh:73-91.  rsqrtf(len2) is restated as 1/sqrtf(len2).
inline void createOrthoNormalBasis(const V3 &N, V3 &T, V3 &B) {
    float len2 = dot(N, N);
    if (len2 < 1e-20f) {
        T = V3(1.0f, 0.0f, 0.0f);
        B = V3(0.0f, 1.0f, 0.0f);
        return;
    }
    V3 Nn = N * (1.0f / sqrtf(len2));
    float s = copysignf(1.0f, Nn.z);
    float a = -1.0f / (s + Nn.z);
    float b = Nn.x * Nn.y * a;
    T = V3(1.0f + s * Nn.x * Nn.x * a, s * b, -s * Nn.x);
    B = cross(Nn, T);
}
// sampling.cuh:105-120
inline V3 sample_cone_direction(Xorwow &rng, const V3 &cone_dir, float cos_theta_max) {
    float u1 = xorwow_uniform(rng);
    float u2 = xorwow_uniform(rng);
    float cos_theta = 1.0f - u1 * (1.0f - cos_theta_max);
    float sin_theta = sqrtf(dm_max(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = TWO_PI_F * u2;
    float sp, cp;
    dm_sincos(phi, &sp, &cp);
    V3 sample_dir(sin_theta * cp, sin_theta * sp, cos_theta);
    V3 T, B;
    createOrthoNormalBasis(cone_dir, T, B);
    return sample_dir.x * T + sample_dir.y * B + sample_dir.z * cone_dir;
}
// sampling.cuh:141-147
inline V3 sample_cosine_hemisphere(Xorwow &rng) {
    float u1 = xorwow_uniform(rng);
    float u2 = xorwow_uniform(rng);
    float r = sqrtf(u1);
    float phi = TWO_PI_F * u2;
    float sp, cp;
    dm_sincos(phi, &sp, &cp);
    return V3(r * cp, r * sp, sqrtf(dm_max(0.0f, 1.0f - u1)));
}
// sampling.cuh:159-164
inline V3 hemisphere_to_world(const V3 &sample, const V3 &N) {
    V3 T, B;
    createOrthoNormalBasis(N, T, B);
    return sample.x * T + sample.y * B + sample.z * N;
}
// sampling.cuh:187-208
inline V3 importance_sample_ggx(Xorwow &rng, const V3 &N, float roughness) {
    float a = roughness * roughness;
    float a2 = a * a;
    float u1 = xorwow_uniform(rng);
    float u2 = xorwow_uniform(rng);
    u2 = dm_min(u2, 0.9999999f);
    float phi = TWO_PI_F * u1;
    float cosTheta = sqrtf((1.0f - u2) / (1.0f + (a2 - 1.0f) * u2));
    float sinTheta = sqrtf(dm_max(0.0f, 1.0f - cosTheta * cosTheta));
    float sp, cp;
    dm_sincos(phi, &sp, &cp);
    V3 H;
    H.x = sinTheta * cp;
    H.y = sinTheta * sp;
    H.z = cosTheta;
    return hemisphere_to_world(H, N);
}

// ---------------------------------------------------------------------------
// PBR helpers (rendering/pbr_utils.cuh, render_utils.cuh)
// ---------------------------------------------------------------------------
inline V3 fresnelSchlick(float cosTheta, const V3 &F0) { // pbr_utils.cuh:16-22
    cosTheta = clamp01(cosTheta);
    float f = 1.0f - cosTheta;
    float f2 = f * f;
    float f5 = f2 * f2 * f;
    return F0 + (V3(1.0f) - F0) * f5;
}
inline float distributionGGX(const V3 &N, const V3 &H, float roughness) { // pbr_utils.cuh:36-47
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH = dm_max(dot(N, H), 0.0f);
    float NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = PI_F * denom * denom;
    return a2 / dm_max(denom, 1e-6f);
}
inline float geometrySchlickGGX(float NdotV, float roughness) { // pbr_utils.cuh:55-61
    float r = (roughness + 1.0f);
    float k = (r * r) * 0.125f;
    return NdotV / (NdotV * (1.0f - k) + k + 1e-6f);
}
inline float geometrySmith(const V3 &N, const V3 &V, const V3 &L, float roughness) { // pbr_utils.cuh:63-71
    float NdotV = dm_max(dot(N, V), 0.0f);
    float NdotL = dm_max(dot(N, L), 0.0f);
    float ggx2 = geometrySchlickGGX(NdotV, roughness);
    float ggx1 = geometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}
inline float geometrySmithTransmission(const V3 &N, const V3 &V, const V3 &L, float roughness) { // path_logic.cuh:33-42
    float NdotV = dm_max(dot(N, V), 0.0f);
    float NdotL = fabsf(dot(N, L));
    float ggx2 = geometrySchlickGGX(NdotV, roughness);
    float ggx1 = geometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}
// pbr_utils.cuh:85-125
inline V3 calculateIridescence(float thickness, float cosTheta, float filmIOR, float baseIOR) {
    cosTheta = clamp01(cosTheta);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sinThetaFilm = sinTheta / filmIOR;
    if (sinThetaFilm * sinThetaFilm > 1.0f)
        return V3(1.0f);
    float cosThetaFilm = sqrtf(1.0f - sinThetaFilm * sinThetaFilm);
    float OPD = 2.0f * filmIOR * thickness * cosThetaFilm;
    float R_air_film_s = (1.0f - filmIOR) / (1.0f + filmIOR);
    R_air_film_s *= R_air_film_s;
    float R_film_base_s = (filmIOR - baseIOR) / (filmIOR + baseIOR);
    R_film_base_s *= R_film_base_s;
    const float inv_wavelengths[3] = {1.0f / 650.0f, 1.0f / 550.0f, 1.0f / 450.0f};
    float result[3];
    float sqrtR1R2 = sqrtf(R_air_film_s * R_film_base_s);
    float R_max = (sqrtf(R_air_film_s) + sqrtf(R_film_base_s));
    R_max *= R_max;
    float inv_R_max = 1.0f / (R_max + 1e-6f);
    for (int i = 0; i < 3; ++i) {
        float delta = TWO_PI_F * OPD * inv_wavelengths[i];
        float R_total = R_air_film_s + R_film_base_s + 2.0f * sqrtR1R2 * dm_cos(delta);
        result[i] = clamp01(R_total * inv_R_max);
    }
    return V3(result[0], result[1], result[2]);
}
inline float schlick_dielectric(float cosTheta, float ior_i, float ior_t) { // pbr_utils.cuh:127-137
    cosTheta = clamp01(cosTheta);
    float r0 = (ior_i - ior_t) / (ior_i + ior_t);
    r0 = r0 * r0;
    float f = 1.0f - cosTheta;
    float f2 = f * f;
    float f5 = f2 * f2 * f;
    return r0 + (1.0f - r0) * f5;
}
inline V3 beerLambert(const V3 &ac, float dist) { // pbr_utils.cuh:154-161
    V3 coeff(dm_max(ac.x, 0.0f), dm_max(ac.y, 0.0f), dm_max(ac.z, 0.0f));
    return V3(dm_exp(-coeff.x * dist), dm_exp(-coeff.y * dist), dm_exp(-coeff.z * dist));
}
inline float attenuate(float distance, float range) { // render_utils.cuh:21-24
    float att = range / (range + distance);
    return att * att;
}
// tex2D<float4>(envMap, u, v) for the texture object Scene::loadHDRI creates (scene.cuh:1007-1013:
// normalizedCoords, addressMode wrap / clamp, cudaFilterModeLinear, element type), restated from the
// CUDA C Programming Guide's "Texture Fetching" appendix: wrap takes frac(u); clamp takes v into
// [0, 1 - 1/H]; xB = x*N - 0.5, i = floor(xB), alpha = frac(xB) kept in 1.8 fixed point (8
// fractional bits; rounding to nearest is OUR choice, the guide does not say -- parity unpinned);
// out-of-range texel indices follow the address mode; the four texels are blended in fp32.
struct EnvMap {
    const float *rgba;
    int w, h;
};
inline V3 env_texel(const EnvMap &E, int i, int j) {
    i %= E.w;
    if (i < 0)
        i += E.w;
    j = j < 0 ? 0 : (j > E.h - 1 ? E.h - 1 : j);
    const float *p = E.rgba + ((size_t)j * E.w + i) * 4;
    return V3(p[0], p[1], p[2]);
}
inline V3 tex2D_env(const EnvMap &E, float u, float v) {
    const float uw = u - floorf(u);
    const float vmax = 1.0f - 1.0f / (float)E.h;
    const float vc = v < 0.0f ? 0.0f : (v >= 1.0f ? vmax : v);
    const float xB = uw * (float)E.w - 0.5f, yB = vc * (float)E.h - 0.5f;
    const float fi = floorf(xB), fj = floorf(yB);
    const float a = rintf((xB - fi) * 256.0f) * (1.0f / 256.0f), b = rintf((yB - fj) * 256.0f) * (1.0f / 256.0f);
    const int i = (int)fi, j = (int)fj;
    const V3 t00 = env_texel(E, i, j), t10 = env_texel(E, i + 1, j), t01 = env_texel(E, i, j + 1),
             t11 = env_texel(E, i + 1, j + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return ((t00 * w00 + t10 * w10) + t01 * w01) + t11 * w11;
}

inline V3 sampleSky(const Ray &r, const V3 &top, const V3 &bottom, bool useSky, const EnvMap &env) { // render_utils.cuh:115-137
    if (!useSky)
        return V3(0.0f);
    if (!env.rgba) {
        float t = 0.5f * (r.dir.y + 1.0f);
        return lerp(bottom, top, t);
    }
    const V3 dir = r.dir;
    float phi = dm_atan2(dir.z, dir.x);
    float theta = dm_acos(dm_max(-1.0f, dm_min(1.0f, dir.y)));
    float u = (phi + PI_F) * (1.0f / TWO_PI_F);
    float v = theta * (1.0f / PI_F);
    return tex2D_env(env, u, v);
}

// ---------------------------------------------------------------------------
// materials (rendering/path_logic.cuh:73-122)
// ---------------------------------------------------------------------------
struct MaterialProps {
    V3 albedo, specular, emission;
    float metallic, roughness, transmission, ior, transmissionRoughness, clearcoat, clearcoatRoughness,
        iridescence, iridescenceThickness, sheen;
    V3 sheenTint;
    void load(const ptrt_materials &m, int id) {
        albedo = m.albedo[id];
        specular = m.specular[id];
        emission = m.emission[id];
        metallic = m.metallic[id];
        roughness = m.roughness[id];
        transmission = m.transmission[id];
        ior = m.ior[id];
        transmissionRoughness = m.transmission_roughness[id];
        clearcoat = m.clearcoat[id];
        clearcoatRoughness = m.clearcoat_roughness[id];
        iridescence = m.iridescence[id];
        iridescenceThickness = m.iridescence_thickness[id];
        sheen = m.sheen[id];
        sheenTint = m.sheen_tint[id];
    }
};

inline V3 clamp_vector_soft(const V3 &v, float max_lum) { // path_logic.cuh:44-52
    float lum = 0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z;
    if (lum > max_lum && lum > 0.0f) {
        float scale = max_lum / lum;
        return v * scale;
    }
    return v;
}

// path_logic.cuh:157-250
V3 evaluateBSDF(const HitInfo &hit, const MaterialProps &mat, const V3 &L, const V3 &V) {
    const V3 N = hit.normal;
    const float NdotV = dm_max(dot(N, V), 0.0f);
    if (NdotV <= 0.0f)
        return V3(0.0f);
    const float metal = clamp01(mat.metallic);
    const float rough = dm_max(mat.roughness, 0.02f);
    const float trans = clamp01(mat.transmission);
    const V3 albedo = mat.albedo;
    V3 specular = mat.specular;
    V3 F0_base = lerp(specular, albedo, metal);
    const float iridescence = clamp01(mat.iridescence);
    if (iridescence > 0.0f) {
        V3 ic = calculateIridescence(mat.iridescenceThickness, NdotV, 1.3f, mat.ior);
        F0_base = lerp(F0_base, ic, iridescence);
    }
    if (trans > 0.0f && metal < 0.1f) {
        const float ior = mat.ior;
        const float transRough = dm_max(mat.transmissionRoughness, rough);
        float ior_ratio = hit.front_face ? (1.0f / ior) : ior;
        float NdotL = dot(N, L);
        if (NdotL > 0.0f) {
            V3 H = normalize(L + V);
            float VdotH = dm_max(dot(V, H), 0.0f);
            float D = distributionGGX(N, H, rough);
            float G = geometrySmith(N, V, L, rough);
            V3 F = fresnelSchlick(VdotH, F0_base);
            V3 spec = (D * G * F) / (4.0f * NdotV * NdotL + 1e-6f);
            return spec * NdotL;
        } else {
            float eta = ior_ratio;
            V3 H = normalize(-(V * eta + L));
            if (dot(N, H) < 0.0f)
                H = -H;
            float VdotH = dm_max(dot(V, H), 0.0f);
            float LdotH = fabsf(dot(L, H));
            float NdotL_abs = fabsf(NdotL);
            float k = 1.0f - eta * eta * (1.0f - VdotH * VdotH);
            if (k < 0.0f)
                return V3(0.0f);
            float D = distributionGGX(N, H, transRough);
            float G = geometrySmithTransmission(N, V, L, transRough);
            V3 F_fresnel = fresnelSchlick(VdotH, F0_base);
            V3 F = V3(1.0f) - F_fresnel;
            float numerator = (eta * eta * (1.0f - metal) * G * D * VdotH * LdotH);
            float pw = eta * VdotH + LdotH;
            float denominator = NdotV * NdotL_abs * (pw * pw); // powf(.,2.0f)
            V3 btdf = (albedo * F * numerator) / (denominator + 1e-6f);
            return btdf * NdotL_abs;
        }
    }
    float NdotL = dm_max(dot(N, L), 0.0f);
    if (NdotL <= 0.0f)
        return V3(0.0f);
    V3 H = normalize(L + V);
    float VdotH = dm_max(dot(V, H), 0.0f);
    float D = distributionGGX(N, H, rough);
    float G = geometrySmith(N, V, L, rough);
    V3 F = fresnelSchlick(VdotH, F0_base);
    specular = (D * G * F) / (4.0f * NdotV * NdotL + 0.001f);
    V3 kS = F;
    V3 kD = (V3(1.0f) - kS) * (1.0f - metal);
    V3 diffuse = kD * albedo / PI_F;
    return (diffuse + specular) * NdotL;
}

// ---------------------------------------------------------------------------
// pdfs (math/pdf.cuh)
// ---------------------------------------------------------------------------
inline float mis_weight(float pdf1, float pdf2) { // pdf.cuh:26-30
    float a = pdf1 * pdf1;
    float b = pdf2 * pdf2;
    return a / (a + b + 1e-10f);
}
inline float pdf_cosine_hemisphere(const V3 &N, const V3 &L) { // pdf.cuh:73-77
    float NdotL = dm_max(dot(N, L), 0.0f);
    return NdotL * (1.0f / PI_F);
}
inline float pdf_ggx_reflect(const V3 &N, const V3 &V, const V3 &L, float roughness) { // pdf.cuh:81-94
    float NdotV = dm_max(dot(N, V), 0.0f);
    if (NdotV == 0.0f)
        return 0.0f;
    V3 H = normalize(V + L);
    float NdotH = dm_max(dot(N, H), 0.0f);
    float VdotH = dm_max(dot(V, H), 0.0f);
    float D = distributionGGX(N, H, roughness);
    float pdf_H = D * NdotH;
    return pdf_H / (4.0f * VdotH + 1e-6f);
}
inline float pdf_ggx_refract(const V3 &N, const V3 &V, const V3 &L, float roughness, float ior_ratio) { // pdf.cuh:97-123
    float NdotV = dm_max(dot(N, V), 0.0f);
    float NdotL = dot(N, L);
    if (NdotV <= 0.0f || NdotL >= 0.0f)
        return 0.0f;
    float eta = ior_ratio;
    V3 H = normalize(-(V * eta + L));
    if (dot(N, H) < 0.0f)
        H = -H;
    float VdotH = dm_max(dot(V, H), 0.0f);
    float LdotH = fabsf(dot(L, H));
    float NdotH = dm_max(dot(N, H), 0.0f);
    float D = distributionGGX(N, H, roughness);
    float pdf_H = D * NdotH;
    float pw = eta * VdotH + LdotH;
    float dwh_dwo = (eta * eta * LdotH) / (pw * pw);
    return pdf_H * fabsf(dwh_dwo);
}
// pdf.cuh:127-220
float material_pdf(const HitInfo &hit, const MaterialProps &mat, const V3 &V, const V3 &L) {
    const V3 N = hit.normal;
    const float NdotV = dm_max(dot(N, V), 0.0f);
    const float NdotL = dm_max(dot(N, L), 0.0f);
    if (NdotV == 0.0f)
        return 0.0f;
    const float metal = clamp01(mat.metallic);
    const float rough = dm_max(mat.roughness, 0.02f);
    const float trans = clamp01(mat.transmission);
    V3 F0_base = lerp(mat.specular, mat.albedo, metal);
    const float iridescence = clamp01(mat.iridescence);
    if (iridescence > 0.0f) {
        V3 ic = calculateIridescence(mat.iridescenceThickness, NdotV, 1.3f, mat.ior);
        F0_base = lerp(F0_base, ic, iridescence);
    }
    V3 F_base = fresnelSchlick(NdotV, F0_base);
    float total_pdf = 0.0f;
    float prob_base = 1.0f;
    const float clearcoat = clamp01(mat.clearcoat);
    if (clearcoat > 0.0f) {
        const float clearcoatRough = dm_max(mat.clearcoatRoughness, 0.001f);
        const V3 F_coat = fresnelSchlick(NdotV, V3(0.04f));
        float F_coat_avg = (F_coat.x + F_coat.y + F_coat.z) * (1.0f / 3.0f);
        float P_coat = clamp01(F_coat_avg * clearcoat);
        if (NdotL > 0.0f)
            total_pdf += P_coat * pdf_ggx_reflect(N, V, L, clearcoatRough);
        prob_base = (1.0f - P_coat);
    }
    if (trans > 0.0f && metal < 0.1f) {
        const float ior = mat.ior;
        const float transRough = dm_max(mat.transmissionRoughness, rough);
        float ior_ratio = hit.front_face ? (1.0f / ior) : ior;
        float reflect_prob = schlick_dielectric(NdotV, 1.0f, ior_ratio); // schlick_dielectric_oneIOR
        if (NdotL > 0.0f) {
            float pdf_reflect = pdf_ggx_reflect(N, V, L, rough);
            total_pdf += prob_base * reflect_prob * pdf_reflect;
            V3 H = normalize(V + L);
            float VdotH = dm_max(dot(V, H), 0.0f);
            float k = 1.0f - ior_ratio * ior_ratio * (1.0f - VdotH * VdotH);
            if (k < 0.0f) {
                float p2 = pdf_ggx_reflect(N, V, L, transRough);
                total_pdf += prob_base * (1.0f - reflect_prob) * p2;
            }
        } else {
            float pdf_refract = pdf_ggx_refract(N, V, L, transRough, ior_ratio);
            total_pdf += prob_base * (1.0f - reflect_prob) * pdf_refract;
        }
        return total_pdf;
    }
    if (NdotL > 0.0f) {
        float max_fresnel = dm_max(F_base.x, dm_max(F_base.y, F_base.z));
        float specular_prob = (metal > 0.0f) ? 1.0f : max_fresnel;
        float pdf_spec = pdf_ggx_reflect(N, V, L, rough);
        float pdf_diffuse = pdf_cosine_hemisphere(N, L);
        total_pdf += prob_base * (specular_prob * pdf_spec + (1.0f - specular_prob) * pdf_diffuse);
    }
    return total_pdf;
}

// ---------------------------------------------------------------------------
// next-event estimation (path_logic.cuh:305-393)
// ---------------------------------------------------------------------------
V3 sample_direct_lighting_with_mat(const HitInfo &hit, const MaterialProps &mat, const Ray &ray_in,
                                   const ptrt_scene_desc &S, Xorwow &rng, V3 &out_L, float &out_pdf,
                                   Counters &cnt) {
    const int nLights = S.light_count;
    if (nLights == 0) {
        out_L = V3(0.0f);
        out_pdf = 0.0f;
        return V3(0.0f);
    }
    V3 direct_light(0.0f);
    const V3 V = -ray_in.dir;
    float r = xorwow_uniform(rng);
    r = dm_min(r, 0.99999994f);
    int light_index = (int)(r * nLights);
    const ptrt_light &light = S.lights[light_index];
    float pdf_pick = 1.0f / (float)nLights;
    V3 L;
    float attenuation = 1.0f;
    float light_dist = 1e30f;
    V3 light_radiance = V3(light.color) * light.intensity;
    float pdf_sample = 1.0f;
    if (light.type == PTRT_LIGHT_DIRECTIONAL) {
        L = -V3(light.direction);
        pdf_sample = pdf_pick;
    } else {
        V3 toLight = V3(light.position) - hit.point;
        float light_dist_sq = length_squared(toLight);
        light_dist = sqrtf(light_dist_sq);
        if (light.radius <= 0.0f) {
            L = toLight / light_dist;
            pdf_sample = pdf_pick;
        } else {
            float sin_theta_max_sq = (light.radius * light.radius) / light_dist_sq;
            sin_theta_max_sq = dm_min(sin_theta_max_sq, 0.9999f);
            float cos_theta_max = sqrtf(1.0f - sin_theta_max_sq);
            L = sample_cone_direction(rng, toLight / light_dist, cos_theta_max);
            float solid_angle = TWO_PI_F * (1.0f - cos_theta_max);
            pdf_sample = (solid_angle > 1e-6f) ? (pdf_pick / solid_angle) : pdf_pick;
        }
        attenuation = attenuate(light_dist, light.range);
        if (light.type == PTRT_LIGHT_SPOT) {
            float theta = dot(L, -V3(light.direction));
            float epsilon = light.inner_cone - light.outer_cone;
            float spotIntensity;
            if (epsilon <= 1e-6f) {
                spotIntensity = (theta >= light.outer_cone) ? 1.0f : 0.0f;
            } else {
                spotIntensity = clampf((theta - light.outer_cone) / epsilon, 0.0f, 1.0f);
            }
            attenuation *= spotIntensity;
        }
    }
    out_L = L;
    out_pdf = pdf_sample;
    V3 shadow_offset = dot(hit.normal, L) > 0.0f ? hit.normal * 1e-4f : -hit.normal * 1e-4f;
    Ray shadowRay(hit.point + shadow_offset, L);
    bool inShadow = bvh_any_hit_tlas(shadowRay, light_dist - 1e-3f, S, cnt);
    { // the same expression as below, evaluated for EVERY sample, only to count the ones that add nothing either way
        bool nonzero = false;
        if (pdf_sample > 0.0f) {
            V3 d0 = clamp_vector_soft(evaluateBSDF(hit, mat, L, V) * light_radiance * attenuation / pdf_sample, 500.0f);
            nonzero = d0.x > 0.0f || d0.y > 0.0f || d0.z > 0.0f;
        }
        if (!nonzero)
            cnt.shadow_zero++;
    }
    if (!inShadow) {
        V3 bsdf = evaluateBSDF(hit, mat, L, V);
        if (pdf_sample > 0.0f) {
            direct_light = bsdf * light_radiance * attenuation / pdf_sample;
            direct_light = clamp_vector_soft(direct_light, 500.0f); // MAX_NEE_CONTRIBUTION
        }
    }
    return direct_light;
}

// ---------------------------------------------------------------------------
// BSDF sampling (path_logic.cuh:490-780)
// ---------------------------------------------------------------------------
bool material_scatter(const HitInfo &hit, const MaterialProps &mat, const Ray &ray_in, Xorwow &rng,
                      V3 &scattered_dir, V3 &attenuation, bool &is_specular_bounce, float &out_pdf) {
    const V3 V = -ray_in.dir;
    const V3 N = hit.normal;
    const float NdotV = dm_max(dot(N, V), 0.0f);
    const float metal = clamp01(mat.metallic);
    const float rough = dm_max(mat.roughness, 0.02f);
    const float trans = clamp01(mat.transmission);
    const V3 albedo = mat.albedo;
    const V3 specular = mat.specular;
    V3 F0_base = lerp(specular, albedo, metal);
    const float iridescence = clamp01(mat.iridescence);
    if (iridescence > 0.0f) {
        V3 ic = calculateIridescence(mat.iridescenceThickness, NdotV, 1.3f, mat.ior);
        F0_base = lerp(F0_base, ic, iridescence);
    }
    V3 F_base_for_NdotV = fresnelSchlick(NdotV, F0_base);
    const float clearcoat = clamp01(mat.clearcoat);
    float P_coat = 0.0f;
    float prob_base = 1.0f;
    float clearcoatRough = 0.0f;
    V3 F0_coat = V3(0.0f);
    if (clearcoat > 0.0f) {
        clearcoatRough = dm_max(mat.clearcoatRoughness, 0.001f);
        F0_coat = V3(0.04f);
        V3 F_coat = fresnelSchlick(NdotV, F0_coat);
        float F_coat_avg = (F_coat.x + F_coat.y + F_coat.z) * (1.0f / 3.0f);
        P_coat = clamp01(F_coat_avg * clearcoat);
        prob_base = (1.0f - P_coat);
    }

    if (trans > 0.0f && metal < 0.1f) {
        const float ior = mat.ior;
        const float transRough = dm_max(mat.transmissionRoughness, rough);
        float ior_ratio = hit.front_face ? (1.0f / ior) : ior;
        float ior_incident = hit.front_face ? 1.0f : ior;
        float ior_transmitted = hit.front_face ? ior : 1.0f;
        float reflect_prob = schlick_dielectric(NdotV, ior_incident, ior_transmitted);
        float refract_prob = 1.0f - reflect_prob;
        float P_trans_reflect = prob_base * reflect_prob;
        float P_trans_refract = prob_base * refract_prob;
        float u = xorwow_uniform(rng);
        V3 H;
        bool is_refraction = false;
        float sample_roughness;
        float eta = ior_ratio;
        if (u < P_coat) {
            sample_roughness = clearcoatRough;
            H = importance_sample_ggx(rng, N, sample_roughness);
            scattered_dir = reflectVec(-V, H);
            is_specular_bounce = (sample_roughness < 0.1f);
        } else if (u < P_coat + P_trans_reflect) {
            sample_roughness = rough;
            H = importance_sample_ggx(rng, N, sample_roughness);
            scattered_dir = reflectVec(-V, H);
            is_specular_bounce = (sample_roughness < 0.1f);
        } else {
            sample_roughness = transRough;
            H = importance_sample_ggx(rng, N, sample_roughness);
            is_refraction = true;
            is_specular_bounce = (sample_roughness < 0.1f);
            float VdotH_tir = dot(V, H);
            if (VdotH_tir < 0.0f)
                H = -H;
            VdotH_tir = fabsf(dot(V, H));
            float k = 1.0f - eta * eta * (1.0f - VdotH_tir * VdotH_tir);
            if (k < 0.0f) {
                scattered_dir = reflectVec(-V, H);
                is_specular_bounce = true;
            } else {
                float cos_t = sqrtf(k);
                scattered_dir = normalize(eta * (-V) + (eta * VdotH_tir - cos_t) * H);
            }
        }
        float NdotL = dot(N, scattered_dir);
        V3 f_total(0.0f);
        float pdf_total = 0.0f;
        V3 F_coat_atten;
        if (is_refraction) {
            V3 Hb = normalize(eta * V + scattered_dir);
            float vh = dm_max(dot(V, Hb), 0.0f);
            F_coat_atten = fresnelSchlick(vh, F0_coat);
        } else {
            V3 Hb = normalize(V + scattered_dir);
            float vh = dm_max(dot(V, Hb), 0.0f);
            F_coat_atten = fresnelSchlick(vh, F0_coat);
        }
        V3 base_attenuation = V3(1.0f) - clearcoat * F_coat_atten;
        if (P_coat > 0.0f && NdotL > 0.0f) {
            V3 H_coat = normalize(V + scattered_dir);
            float NdotH_coat = dm_max(dot(N, H_coat), 0.0f);
            float VdotH_coat = dm_max(dot(V, H_coat), 0.0f);
            float D_coat = distributionGGX(N, H_coat, clearcoatRough);
            float G_coat = geometrySmith(N, V, scattered_dir, clearcoatRough);
            V3 F_coat = fresnelSchlick(VdotH_coat, F0_coat);
            float pdf_L_coat = (D_coat * NdotH_coat) / (4.0f * VdotH_coat + 1e-6f);
            pdf_total += P_coat * pdf_L_coat;
            V3 brdf_coat = (D_coat * G_coat * F_coat) / (4.0f * NdotV * NdotL + 1e-6f);
            f_total = f_total + clearcoat * brdf_coat * NdotL;
        }
        if (P_trans_reflect > 0.0f && NdotL > 0.0f) {
            V3 H_refl = normalize(V + scattered_dir);
            float NdotH_refl = dm_max(dot(N, H_refl), 0.0f);
            float VdotH_refl = dm_max(dot(V, H_refl), 0.0f);
            float D_refl = distributionGGX(N, H_refl, rough);
            float G_refl = geometrySmith(N, V, scattered_dir, rough);
            V3 F_refl = fresnelSchlick(VdotH_refl, F0_base);
            float pdf_L_refl = (D_refl * NdotH_refl) / (4.0f * VdotH_refl + 1e-6f);
            pdf_total += P_trans_reflect * pdf_L_refl;
            V3 brdf_refl = (D_refl * G_refl * F_refl) / (4.0f * NdotV * NdotL + 1e-6f);
            f_total = f_total + brdf_refl * NdotL * base_attenuation;
        }
        if (P_trans_refract > 0.0f && NdotL < 0.0f) {
            V3 H_refr = normalize(-(V * eta + scattered_dir));
            if (dot(N, H_refr) < 0.0f)
                H_refr = -H_refr;
            float VdotH_refr = dm_max(dot(V, H_refr), 0.0f);
            float LdotH_refr = fabsf(dot(scattered_dir, H_refr));
            float NdotH_refr = dm_max(dot(N, H_refr), 0.0f);
            float NdotL_abs = fabsf(NdotL);
            float k = 1.0f - eta * eta * (1.0f - VdotH_refr * VdotH_refr);
            if (k >= 0.0f) {
                float D_refr = distributionGGX(N, H_refr, transRough);
                float G_refr = geometrySmithTransmission(N, V, scattered_dir, transRough);
                float pw = eta * VdotH_refr + LdotH_refr;
                float dwh_dwo = (eta * eta * LdotH_refr) / (pw * pw);
                float pdf_L_refr = (D_refr * NdotH_refr * fabsf(dwh_dwo));
                pdf_total += P_trans_refract * pdf_L_refr;
                V3 F_refr = V3(1.0f) - fresnelSchlick(VdotH_refr, F0_base);
                float numerator = (eta * eta * (1.0f - metal) * G_refr * D_refr * VdotH_refr * LdotH_refr);
                float denominator = NdotV * NdotL_abs * (pw * pw);
                V3 btdf = (albedo * F_refr * numerator) / (denominator + 1e-6f);
                f_total = f_total + btdf * NdotL_abs * base_attenuation;
            }
        }
        if (is_refraction && NdotL > 0.0f) {
            V3 H_refl = normalize(V + scattered_dir);
            float NdotH_refl = dm_max(dot(N, H_refl), 0.0f);
            float VdotH_refl = dm_max(dot(V, H_refl), 0.0f);
            float D_refl = distributionGGX(N, H_refl, transRough);
            float G_refl = geometrySmith(N, V, scattered_dir, transRough);
            float pdf_L_refl = (D_refl * NdotH_refl) / (4.0f * VdotH_refl + 1e-6f);
            pdf_total += P_trans_refract * pdf_L_refl;
            V3 brdf_refl = (D_refl * G_refl * V3(1.0f)) / (4.0f * NdotV * NdotL + 1e-6f);
            f_total = f_total + brdf_refl * NdotL * base_attenuation;
        }
        out_pdf = dm_max(pdf_total, 1e-6f);
        attenuation = f_total / out_pdf;
        return true;
    }

    float max_fresnel = dm_max(F_base_for_NdotV.x, dm_max(F_base_for_NdotV.y, F_base_for_NdotV.z));
    float specular_prob = (metal > 0.0f) ? 1.0f : max_fresnel;
    float P_opaque_spec = prob_base * specular_prob;
    float P_opaque_diff = prob_base * (1.0f - specular_prob);
    float u = xorwow_uniform(rng);
    if (u < P_coat) {
        V3 H = importance_sample_ggx(rng, N, clearcoatRough);
        scattered_dir = reflectVec(-V, H);
        is_specular_bounce = (clearcoatRough < 0.1f);
    } else if (u < P_coat + P_opaque_spec) {
        V3 H = importance_sample_ggx(rng, N, rough);
        scattered_dir = reflectVec(-V, H);
        is_specular_bounce = (rough < 0.1f);
    } else if (P_opaque_diff > 1e-6f) {
        V3 hemi = sample_cosine_hemisphere(rng);
        scattered_dir = hemisphere_to_world(hemi, N);
        is_specular_bounce = false;
    } else {
        return false;
    }
    scattered_dir = normalize(scattered_dir);
    float NdotL = dm_max(dot(N, scattered_dir), 0.0f);
    V3 f_total(0.0f);
    float pdf_total = 0.0f;
    if (P_coat > 0.0f) {
        V3 H_coat = normalize(V + scattered_dir);
        float NdotH_coat = dm_max(dot(N, H_coat), 0.0f);
        float VdotH_coat = dm_max(dot(V, H_coat), 0.0f);
        float D_coat = distributionGGX(N, H_coat, clearcoatRough);
        float G_coat = geometrySmith(N, V, scattered_dir, clearcoatRough);
        V3 F_coat = fresnelSchlick(VdotH_coat, F0_coat);
        float pdf_L_coat = (D_coat * NdotH_coat) / (4.0f * VdotH_coat + 1e-6f);
        pdf_total += P_coat * pdf_L_coat;
        V3 brdf_coat = (D_coat * G_coat * F_coat) / (4.0f * NdotV * NdotL + 1e-6f);
        f_total = f_total + clearcoat * brdf_coat * NdotL;
    }
    V3 H_for_base = normalize(V + scattered_dir);
    float VdotH_for_base = dm_max(dot(V, H_for_base), 0.0f);
    V3 F_coat_atten = fresnelSchlick(VdotH_for_base, F0_coat);
    V3 base_attenuation = V3(1.0f) - clearcoat * F_coat_atten;
    V3 H_spec = H_for_base;
    float NdotH_spec = dm_max(dot(N, H_spec), 0.0f);
    float VdotH_spec = VdotH_for_base;
    float D_spec = distributionGGX(N, H_spec, rough);
    float G_spec = geometrySmith(N, V, scattered_dir, rough);
    V3 F_spec = fresnelSchlick(VdotH_spec, F0_base);
    float pdf_L_spec = (D_spec * NdotH_spec) / (4.0f * VdotH_spec + 1e-6f);
    pdf_total += P_opaque_spec * pdf_L_spec;
    V3 brdf_spec = (D_spec * G_spec * F_spec) / (4.0f * NdotV * NdotL + 1e-6f);
    f_total = f_total + brdf_spec * NdotL * base_attenuation;
    if (P_opaque_diff > 1e-6f) {
        float pdf_L_diff = NdotL / PI_F;
        pdf_total += P_opaque_diff * pdf_L_diff;
        const float sheen = clamp01(mat.sheen);
        V3 kD = (V3(1.0f) - F_base_for_NdotV) * (1.0f - metal);
        V3 f_diff = (kD * albedo / PI_F) * NdotL;
        if (sheen > 0.0f) {
            float FH = 1.0f - dm_max(dot(V, H_for_base), 0.0f);
            float FH5 = FH * FH * FH * FH * FH;
            V3 Csheen = lerp(V3(1.0f), mat.sheenTint, 0.5f);
            f_diff = f_diff + sheen * Csheen * FH5 * NdotL;
        }
        f_total = f_total + f_diff * base_attenuation;
    }
    out_pdf = pdf_total;
    attenuation = f_total / dm_max(pdf_total, 1e-6f);
    return true;
}

// ---------------------------------------------------------------------------
// the integrator (path_logic.cuh:782-899)
// ---------------------------------------------------------------------------
V3 tracePath(Ray ray, const ptrt_scene_desc &S, Xorwow &rng, int max_depth, V3 &out_first_normal,
             float &out_first_depth, int &out_first_objectId, Counters &cnt) {
    V3 accumulated_color(0.0f);
    V3 throughput(1.0f);
    bool prev_was_specular = true;
    const V3 skyTop = S.sky_top, skyBottom = S.sky_bottom;
    for (int bounce = 0; bounce < max_depth; ++bounce) {
        HitInfo hit = traceRay(ray, S, cnt);
        if (bounce == 0) {
            if (!hit.hit) {
                out_first_normal = V3(0.0f);
                out_first_depth = 1e30f;
                out_first_objectId = -1;
            } else {
                out_first_normal = hit.normal;
                out_first_depth = hit.t;
                out_first_objectId = hit.mesh_index;
            }
        }
        if (!hit.hit) {
            V3 sky = sampleSky(ray, skyTop, skyBottom, S.use_sky != 0, EnvMap{S.env_rgba, S.env_width, S.env_height});
            accumulated_color = accumulated_color + throughput * sky;
            break;
        }
        MaterialProps mat;
        mat.load(S.materials, hit.mesh_index);
        const V3 V = -ray.dir;
        if (!hit.front_face) {
            V3 T_unit(dm_max(1e-6f, mat.albedo.x), dm_max(1e-6f, mat.albedo.y), dm_max(1e-6f, mat.albedo.z));
            V3 absorption(-dm_log(T_unit.x), -dm_log(T_unit.y), -dm_log(T_unit.z));
            throughput = throughput * beerLambert(absorption, hit.t);
        }
        V3 emission = mat.emission;
        if (emission.x > 0.0f || emission.y > 0.0f || emission.z > 0.0f) {
            if (bounce == 0 || prev_was_specular)
                accumulated_color = accumulated_color + throughput * emission;
        }
        if (!ray.spec) {
            V3 L_nee;
            float pdf_nee;
            V3 brdf_nee = sample_direct_lighting_with_mat(hit, mat, ray, S, rng, L_nee, pdf_nee, cnt);
            if (brdf_nee.x > 0.0f || brdf_nee.y > 0.0f || brdf_nee.z > 0.0f) {
                if (pdf_nee > 0.0f) {
                    float pdf_brdf = material_pdf(hit, mat, V, L_nee);
                    float w = mis_weight(pdf_nee, pdf_brdf);
                    accumulated_color = accumulated_color + throughput * brdf_nee * w;
                }
            }
        }
        V3 scatter_dir, attenuation;
        bool is_specular;
        float pdf_brdf;
        if (!material_scatter(hit, mat, ray, rng, scatter_dir, attenuation, is_specular, pdf_brdf))
            break;
        prev_was_specular = is_specular;
        if (bounce >= 2) { // RUSSIAN_ROULETTE_START_BOUNCE, path_logic.cuh:24
            float p = dm_max(0.05f, dm_min(0.95f, dm_max(throughput.x, dm_max(throughput.y, throughput.z))));
            if (xorwow_uniform(rng) > p)
                break;
            throughput = throughput / p;
        }
        throughput = throughput * attenuation;
        throughput = clamp_vector_soft(throughput, 50.0f); // MAX_BOUNCE_WEIGHT
        V3 offset_origin;
        if (dot(scatter_dir, hit.normal) > 0.0f)
            offset_origin = hit.point + hit.normal * 1e-4f;
        else
            offset_origin = hit.point - hit.normal * 1e-4f;
        ray = Ray(offset_origin, scatter_dir, is_specular);
    }
    accumulated_color = clamp_vector_soft(accumulated_color, 100.0f); // MAX_FINAL_RADIANCE
    return accumulated_color;
}

// ---------------------------------------------------------------------------
// primary rays and jitter
// ---------------------------------------------------------------------------
// taa.cuh:19-36 (entry 15 is {0.0625, 0.592593} as written there)
const float HALTON16[16][2] = {
    {0.500000f, 0.333333f}, {0.250000f, 0.666667f}, {0.750000f, 0.111111f}, {0.125000f, 0.444444f},
    {0.625000f, 0.777778f}, {0.375000f, 0.222222f}, {0.875000f, 0.555556f}, {0.062500f, 0.888889f},
    {0.562500f, 0.037037f}, {0.312500f, 0.370370f}, {0.812500f, 0.703704f}, {0.187500f, 0.148148f},
    {0.687500f, 0.481481f}, {0.437500f, 0.814815f}, {0.937500f, 0.259259f}, {0.062500f, 0.592593f},
};
inline void getTAAJitter(int frame_index, float &jx, float &jy) { // taa.cuh:41-61
    int idx = frame_index % 16;
    jx = HALTON16[idx][0] - 0.5f;
    jy = HALTON16[idx][1] - 0.5f;
}
// sampling.cuh:15-43
inline void next_blue_noise(const float *table, int x, int y, int frame, float &ou, float &ov) {
    int bx = x & 63;
    int by = y & 63;
    float val_x = table[(by * 64 + bx) * 2 + 0];
    float val_y = table[(by * 64 + bx) * 2 + 1];
    uint32_t hash = (uint32_t)frame * 0x9e3779b9u;
    hash ^= (hash >> 15);
    hash *= 0x85ebca6bu;
    hash ^= (hash >> 13);
    hash *= 0xc2b2ae35u;
    hash ^= (hash >> 16);
    float shift_x = (hash & 0xFFFFFF) * (1.0f / 16777216.0f);
    hash *= 0x85ebca6bu;
    float shift_y = (hash & 0xFFFFFF) * (1.0f / 16777216.0f);
    float u = val_x + shift_x;
    float v = val_y + shift_y;
    if (u >= 1.0f)
        u -= 1.0f;
    if (v >= 1.0f)
        v -= 1.0f;
    ou = u;
    ov = v;
}
// camera.cuh:23-30.  The two curand_uniform calls are arguments of one constructor;
// this restatement draws x first, then y (the order clang and nvcc evaluate
// braced/ctor arguments for this pattern is left to right).
inline V3 random_in_unit_disk(Xorwow &rng) {
    V3 p;
    do {
        float a = xorwow_uniform(rng);
        float b = xorwow_uniform(rng);
        p = 2.0f * V3(a, b, 0.0f) - V3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}
// camera.cuh:156-166, 201-205
inline Ray camera_get_ray(const ptrt_camera &c, float s, float t, Xorwow &rng) {
    const V3 origin = c.origin, llc = c.lower_left_corner, horizontal = c.horizontal, vertical = c.vertical;
    if (c.lens_radius <= 0) {
        V3 rd = llc + s * horizontal + t * vertical - origin;
        return Ray(origin, normalize(rd), true);
    }
    V3 rd = c.lens_radius * random_in_unit_disk(rng);
    V3 offset = V3(c.u) * rd.x + V3(c.v) * rd.y;
    V3 ray_dir = llc + s * horizontal + t * vertical - origin - offset;
    return Ray(origin + offset, normalize(ray_dir), true);
}

// one pixel of path_trace_kernel (scene_kernels.cuh:130-193)
inline void path_trace_pixel(const ptrt_scene_desc &S, const float *blue_noise, int x, int y, int width,
                             int height, int spp, int max_depth, int frame_count, Xorwow &rng, float *accum3,
                             float *normal3, float *depth, int32_t *objectId, Counters &cnt) {
    V3 avg_color(0.0f);
    V3 first_normal(0.0f);
    float first_depth = 1e30f;
    int first_objectId = -1;
    for (int s = 0; s < spp; ++s) {
        float tjx, tjy, bnx, bny;
        getTAAJitter(frame_count + s, tjx, tjy);
        next_blue_noise(blue_noise, x, y, frame_count + s, bnx, bny);
        float jitter_x = tjx + (bnx - 0.5f) * 0.25f;
        float jitter_y = tjy + (bny - 0.5f) * 0.25f;
        const float u = (x + 0.5f + jitter_x) / width;
        const float v = 1.0f - (y + 0.5f + jitter_y) / height;
        Ray ray = camera_get_ray(S.camera, u, v, rng);
        V3 sample_normal;
        float sample_depth = 0.0f;
        int sample_objectId = -1;
        cnt.paths++;
        V3 sample_color = tracePath(ray, S, rng, max_depth, sample_normal, sample_depth, sample_objectId, cnt);
        avg_color = avg_color + sample_color;
        if (s == 0) {
            first_normal = sample_normal;
            first_depth = sample_depth;
            first_objectId = sample_objectId;
        }
    }
    V3 out = avg_color / (float)spp;
    accum3[0] = out.x; accum3[1] = out.y; accum3[2] = out.z;
    normal3[0] = first_normal.x; normal3[1] = first_normal.y; normal3[2] = first_normal.z;
    *depth = first_depth;
    *objectId = first_objectId;
}

// ---------------------------------------------------------------------------
// tonemap (scene.cuh:2004-2047, render_utils.cuh:77-95)
// ---------------------------------------------------------------------------
inline V3 mat3_mul(const float m[9], const V3 &v) { // matrix.cuh:35-39 (literal, unfused)
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
              m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
inline V3 aces_tonemap(V3 color) {
    const float IN[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    const float OUT[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
    V3 aces = mat3_mul(IN, color);
    V3 a = aces * (aces + 0.0245786f) - 0.000090537f;
    V3 b = aces * (0.983729f * aces + 0.4329510f) + 0.238081f;
    aces = clampv(a / b, 0.0f, 1.0f);
    aces = mat3_mul(OUT, aces);
    return clampv(aces, 0.0f, 1.0f);
}
inline float srgb_oetf(float c) {
    return (c <= 0.0031308f) ? 12.92f * c : 1.055f * dm_pow(c, 1.0f / 2.4f) - 0.055f;
}
inline void tonemap_pixel(const float *in3, int total_samples, uint8_t *out3) {
    if (total_samples == 0) {
        out3[0] = out3[1] = out3[2] = 0;
        return;
    }
    V3 color = V3(in3[0], in3[1], in3[2]) / (float)total_samples;
    color = aces_tonemap(color);
    color.x = srgb_oetf(color.x);
    color.y = srgb_oetf(color.y);
    color.z = srgb_oetf(color.z);
    const V3 rgb = clampv(color, 0.f, 1.f) * 255.99f;
    out3[0] = (uint8_t)rgb.x;
    out3[1] = (uint8_t)rgb.y;
    out3[2] = (uint8_t)rgb.z;
}

template <class F> void parallel_rows(int rows, int threads, F f) {
    if (threads <= 1) {
        for (int r = 0; r < rows; ++r)
            f(r, 0);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (;;) {
                int r = next.fetch_add(1);
                if (r >= rows)
                    break;
                f(r, t);
            }
        });
    for (auto &th : pool)
        th.join();
}

} // namespace

// ===========================================================================
// C entry points of the oracle (used through ctypes by tests/ and bench.py)
// ===========================================================================
extern "C" {

struct oracle_render_args {
    int32_t width, height;        /* full frame                                  */
    int32_t tile_y0, tile_rows;   /* rows rendered (buffers below are tile-sized) */
    int32_t spp, max_depth, frame_count;
    int32_t threads;              /* worker threads (>=1)                        */
    const float *blue_noise;      /* 64*64*2                                     */
    uint32_t *rng;                /* in/out: tile_rows*width*6 {d,v0..v4}         */
    float *accum;                 /* out: tile_rows*width*3                      */
    float *normal;                /* out: tile_rows*width*3                      */
    float *depth;                 /* out: tile_rows*width                        */
    int32_t *object_id;           /* out: tile_rows*width                        */
    uint64_t extension_rays, shadow_rays, paths; /* out */
    uint64_t shadow_rays_walked; /* out: shadow_rays minus the light samples whose value is exactly zero (ptrt.h ptrt_stats) */
};

int oracle_has_fma(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }

/* init_curand_kernel (scene_kernels.cuh:26-35) for global pixel indices
 * [first, first+count): curand_init(seed, idx, 0).  Output {d,v0..v4} per pixel. */
void oracle_xorwow_init(unsigned long long seed, unsigned long long first, unsigned long long count,
                        uint32_t *out) {
    if (count == 0)
        return;
    const GF2Mat &P = subsequence_matrix();
    Xorwow s;
    xorwow_seed(s, seed, CURAND_CONSTANTS);
    xorwow_skip_subsequences(s, first);
    for (unsigned long long i = 0; i < count; ++i) {
        out[i * 6 + 0] = s.d;
        for (int k = 0; k < 5; ++k)
            out[i * 6 + 1 + k] = s.v[k];
        uint32_t nv[5];
        gf2_apply(P, s.v, nv);
        for (int k = 0; k < 5; ++k)
            s.v[k] = nv[k];
    }
}

/* Generator with caller-chosen scrambling constants: lets tests pin the
 * recurrence, the subsequence jump and the state layout against rocRAND's
 * independent XORWOW (which differs from cuRAND only in these four constants). */
void oracle_xorwow_custom(unsigned long long seed, unsigned long long subsequence, uint32_t xor0, uint32_t xor1,
                          uint32_t mul0, uint32_t mul1, int n_draws, uint32_t *out_draws, uint32_t *out_state6) {
    Xorwow s;
    SeedConstants c = {xor0, xor1, mul0, mul1};
    xorwow_seed(s, seed, c);
    xorwow_skip_subsequences(s, subsequence);
    if (out_state6) {
        out_state6[0] = s.d;
        for (int k = 0; k < 5; ++k)
            out_state6[1 + k] = s.v[k];
    }
    for (int i = 0; i < n_draws; ++i)
        out_draws[i] = xorwow_next(s);
}

/* raw draws / uniforms from a given state (advances it) */
void oracle_xorwow_draw(uint32_t *state6, int n, uint32_t *out_u32, float *out_uniform) {
    Xorwow s;
    s.d = state6[0];
    for (int k = 0; k < 5; ++k)
        s.v[k] = state6[1 + k];
    for (int i = 0; i < n; ++i) {
        if (out_u32) {
            out_u32[i] = xorwow_next(s);
        } else {
            out_uniform[i] = xorwow_uniform(s);
        }
    }
    state6[0] = s.d;
    for (int k = 0; k < 5; ++k)
        state6[1 + k] = s.v[k];
}

/* path_trace_kernel over a tile */
int oracle_render(const ptrt_scene_desc *scene, oracle_render_args *a) {
    if (!scene || !a || !a->rng || !a->accum || !a->normal || !a->depth || !a->object_id || !a->blue_noise)
        return -1;
    if (!oracle_has_fma())
        return -2;
    const int W = a->width, H = a->height;
    const int y0 = a->tile_y0, rows = a->tile_rows;
    if (W <= 0 || H <= 0 || y0 < 0 || rows < 0 || y0 + rows > H)
        return -1;
    const int threads = a->threads < 1 ? 1 : a->threads;
    std::vector<Counters> cnt(threads);
    const ptrt_scene_desc &S = *scene;
    parallel_rows(rows, threads, [&](int r, int tid) {
        const int y = y0 + r;
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)r * W + x;
            Xorwow rng;
            rng.d = a->rng[i * 6];
            for (int k = 0; k < 5; ++k)
                rng.v[k] = a->rng[i * 6 + 1 + k];
            if (S.materials.count > 0) { // `if (!materials) return;` scene_kernels.cuh:139
                path_trace_pixel(S, a->blue_noise, x, y, W, H, a->spp, a->max_depth, a->frame_count, rng,
                                 a->accum + i * 3, a->normal + i * 3, a->depth + i, a->object_id + i, cnt[tid]);
            }
            a->rng[i * 6] = rng.d;
            for (int k = 0; k < 5; ++k)
                a->rng[i * 6 + 1 + k] = rng.v[k];
        }
    });
    a->extension_rays = a->shadow_rays = a->paths = a->shadow_rays_walked = 0;
    for (auto &c : cnt) {
        a->extension_rays += c.extension;
        a->shadow_rays += c.shadow;
        a->paths += c.paths;
        a->shadow_rays_walked += c.shadow - c.shadow_zero;
    }
    return 0;
}

/* tonemap_kernel over a tile: in = rows*W*3 floats (top-down), out = rows*W*3
 * bytes, bottom-up within the tile. */
void oracle_tonemap(const float *accum, int width, int rows, int total_samples, uint8_t *out_rgb8, int threads) {
    parallel_rows(rows, threads < 1 ? 1 : threads, [&](int r, int) {
        for (int x = 0; x < width; ++x) {
            const size_t i = (size_t)r * width + x;
            const size_t o = ((size_t)(rows - 1 - r) * width + x) * 3;
            tonemap_pixel(accum + i * 3, total_samples, out_rgb8 + o);
        }
    });
}

/* trace_single_ray_kernel (scene_kernels.cuh:38-49), batched */
void oracle_trace_rays(const ptrt_scene_desc *scene, const float *origins, const float *directions, int n,
                       ptrt_hit *out) {
    Counters cnt;
    for (int i = 0; i < n; ++i) {
        Ray ray(V3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]),
                V3(directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2]));
        HitInfo h = traceRay(ray, *scene, cnt);
        ptrt_hit &o = out[i];
        o.hit = h.hit ? 1 : 0;
        o.t = h.t;
        o.point = {h.point.x, h.point.y, h.point.z};
        o.normal = {h.normal.x, h.normal.y, h.normal.z};
        o.mesh_index = h.mesh_index;
        o.front_face = h.front_face ? 1 : 0;
        o.u = h.u;
        o.v = h.v;
        o.face_index = h.face_index;
        o.local_point = {h.localPoint.x, h.localPoint.y, h.localPoint.z};
    }
}

/* shadow query (bvh_any_hit_tlas), batched: out[i] = 1 if occluded */
void oracle_any_hit(const ptrt_scene_desc *scene, const float *origins, const float *directions, const float *tmax,
                    int n, int32_t *out) {
    Counters cnt;
    for (int i = 0; i < n; ++i) {
        Ray ray(V3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]),
                V3(directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2]));
        out[i] = bvh_any_hit_tlas(ray, tmax[i], *scene, cnt) ? 1 : 0;
    }
}

/* taa.cuh:41-61, exposed so tests can set it beside the reference's own getTAAJitter (oracle/_ref) */
void oracle_taa_jitter(int frame_index, float *out2) { getTAAJitter(frame_index, out2[0], out2[1]); }

/* mat3 * vec3 exactly as aces_tonemap forms it (matrix.cuh:35-39), exposed for the known-answer test against the
 * reference's own mat3 (oracle/_ref, tests/test_ref_probe.py) */
void oracle_mat3_mul(const float *m9, const float *v3, float *out3) {
    const V3 r = mat3_mul(m9, V3(v3[0], v3[1], v3[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}

/* deterministic math, exposed for tests: op 0 sin, 1 cos, 2 exp, 3 log, 4 pow(x,y) */
void oracle_detmath(int op, const float *x, const float *y, int n, float *out) {
    for (int i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = dm_sin(x[i]); break;
        case 1: out[i] = dm_cos(x[i]); break;
        case 2: out[i] = dm_exp(x[i]); break;
        case 3: out[i] = dm_log(x[i]); break;
        case 5: out[i] = dm_atan2(x[i], y[i]); break;
        case 6: out[i] = dm_acos(x[i]); break;
        default: out[i] = dm_pow(x[i], y[i]); break;
        }
    }
}

/* scalar known-answer hooks for the shading functions:
 * evaluates BSDF / pdf / scatter for one material and fixed directions.
 * io: in  N(3) V(3) L(3) front_face(1) ; out f(3) pdf(1) */
void oracle_eval_bsdf(const ptrt_materials *mats, int mat_id, const float *N, const float *V, const float *L,
                      int front_face, float *out_f3, float *out_pdf) {
    MaterialProps m;
    m.load(*mats, mat_id);
    HitInfo h;
    h.hit = true;
    h.normal = V3(N[0], N[1], N[2]);
    h.front_face = front_face != 0;
    V3 f = evaluateBSDF(h, m, V3(L[0], L[1], L[2]), V3(V[0], V[1], V[2]));
    out_f3[0] = f.x; out_f3[1] = f.y; out_f3[2] = f.z;
    *out_pdf = material_pdf(h, m, V3(V[0], V[1], V[2]), V3(L[0], L[1], L[2]));
}

/* one material_scatter call from a given generator state:
 * out8 = dir(3) attenuation(3) pdf(1) flags(1: bit0 ok, bit1 specular) */
void oracle_scatter(const ptrt_materials *mats, int mat_id, const float *N, const float *ray_dir, int front_face,
                    uint32_t *state6, float *out8) {
    MaterialProps m;
    m.load(*mats, mat_id);
    HitInfo h;
    h.hit = true;
    h.normal = V3(N[0], N[1], N[2]);
    h.front_face = front_face != 0;
    Xorwow s;
    s.d = state6[0];
    for (int k = 0; k < 5; ++k)
        s.v[k] = state6[1 + k];
    Ray r(V3(0.0f), V3(ray_dir[0], ray_dir[1], ray_dir[2]));
    V3 dir, att;
    bool spec = false;
    float pdf = 0.0f;
    bool ok = material_scatter(h, m, r, s, dir, att, spec, pdf);
    out8[0] = dir.x; out8[1] = dir.y; out8[2] = dir.z;
    out8[3] = att.x; out8[4] = att.y; out8[5] = att.z;
    out8[6] = pdf;
    out8[7] = (float)((ok ? 1 : 0) | (spec ? 2 : 0));
    state6[0] = s.d;
    for (int k = 0; k < 5; ++k)
        state6[1 + k] = s.v[k];
}

} // extern "C"
