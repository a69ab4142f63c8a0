// ref_probe.cpp -- TEST INFRASTRUCTURE.  Driver TU for oracle/_ref/ref_probe: the part of the
// reference that compiles here from its OWN, unmodified sources, where they lie under
// /root/reference/src, with g++ and the real CUDA runtime headers this image happens to ship
// (triton's bundled copy; nothing is stubbed).  That part is the curand-free headers:
//
//   common/bluenoise.cuh:79-177   BlueNoiseGenerator::generateBlueNoise2D  (the 64x64x2 table)
//   pathtracer/rendering/taa.cuh:19-61   Halton-16 table + getTAAJitter
//   pathtracer/scene/lights.cuh:12-22    Light (layout the C ABI's ptrt_light mirrors)
//   common/vec3.cuh, common/ray.cuh      vec3 / Ray layout, vec3's host-side arithmetic (known answers)
//   common/matrix.cuh:8-90               mat3: the 3x3 products of the ACES tonemap (render_utils.cuh:77-95)
//   common/vec4.cuh:13-122               vec4: layout, constructors, operators (`/` multiplies by the reciprocal), dot, length
//   common/triangle.cuh:15-92            Triangle: the input type of Scene::addTriangles -- layout, constructor (e1, e2, n),
//                                        normal, area, bounds, the two-sided intersect of the old intersection code
// With these the curand-free headers of the reference are exhausted.
//
// Everything else on the path includes <curand_kernel.h> (pathtracer/math/mathutils.cuh:11, pulled
// in by common/mat4.cuh:6), which the image lacks, so it is unbuildable here and stays
// "parity unpinned" (DESIGN.md section 5).  initBlueNoise() (cudaMemcpyToSymbol) is never
// referenced; -ffunction-sections + --gc-sections drops it, so no CUDA library is needed.
//
// Output: one JSON document on stdout (tests/golden/make_ref_probe_golden.py stores it).
#define BLUE_NOISE_IMPLEMENTATION
#include "common/vec3.cuh"
#include "common/ray.cuh"
#include "common/bluenoise.cuh"
#include "common/matrix.cuh"
#include "common/vec4.cuh"
#include "common/triangle.cuh"
#include "pathtracer/rendering/taa.cuh"
#include "pathtracer/scene/lights.cuh"

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint32_t bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char **argv) {
    const char *raw = argc > 1 ? argv[1] : nullptr;  // optional: write the table's raw floats here
    std::vector<float> bn = BlueNoiseGenerator::generateBlueNoise2D(BLUE_NOISE_SIZE, 25);
    if (raw) {
        FILE *f = fopen(raw, "wb");
        if (!f || fwrite(bn.data(), 4, bn.size(), f) != bn.size()) return 2;
        fclose(f);
    }
    printf("{\n \"blue_noise\": {\"size\": %d, \"channels\": %d, \"count\": %zu, \"bits\": [", BLUE_NOISE_SIZE,
           BLUE_NOISE_CHANNELS, bn.size());
    for (size_t i = 0; i < bn.size(); ++i) printf("%s%u", i ? "," : "", bits(bn[i]));
    printf("]},\n \"taa_jitter_bits\": [");
    for (int f = 0; f < 64; ++f) {
        float2 j = getTAAJitter(f);
        printf("%s[%u,%u]", f ? "," : "", bits(j.x), bits(j.y));
    }
    printf("],\n \"taa_sequence_length\": %d,\n", TAA_SEQUENCE_LENGTH);
    Light L;
    printf(" \"layout\": {\"vec3\": %zu, \"Ray\": %zu, \"Light\": %zu, \"Light.type\": %zu, \"Light.position\": %zu, "
           "\"Light.direction\": %zu, \"Light.color\": %zu, \"Light.intensity\": %zu, \"Light.range\": %zu, "
           "\"Light.innerCone\": %zu, \"Light.outerCone\": %zu, \"Light.radius\": %zu},\n",
           sizeof(vec3), sizeof(Ray), sizeof(Light), offsetof(Light, type), offsetof(Light, position),
           offsetof(Light, direction), offsetof(Light, color), offsetof(Light, intensity), offsetof(Light, range),
           offsetof(Light, innerCone), offsetof(Light, outerCone), offsetof(Light, radius));
    printf(" \"light_defaults_bits\": {\"type\": %d, \"position\": [%u,%u,%u], \"direction\": [%u,%u,%u], "
           "\"color\": [%u,%u,%u], \"intensity\": %u, \"range\": %u, \"innerCone\": %u, \"outerCone\": %u, \"radius\": %u},\n",
           (int)L.type, bits(L.position.x), bits(L.position.y), bits(L.position.z), bits(L.direction.x),
           bits(L.direction.y), bits(L.direction.z), bits(L.color.x), bits(L.color.y), bits(L.color.z),
           bits(L.intensity), bits(L.range), bits(L.innerCone), bits(L.outerCone), bits(L.radius));
    printf(" \"light_types\": {\"LIGHT_POINT\": %d, \"LIGHT_DIRECTIONAL\": %d, \"LIGHT_SPOT\": %d},\n",
           (int)LIGHT_POINT, (int)LIGHT_DIRECTIONAL, (int)LIGHT_SPOT);
    // common/vec3.cuh:99-157 as the HOST compiler sees it (the arithmetic the scene-building code runs in:
    // Scene::add*, transforms, the camera frame): 48 seeded input pairs, results as bit patterns.  The kernels'
    // device arithmetic (nvcc's fmad contraction) is a different matter and is NOT what this pins.
    printf(" \"vec3_kat\": [");
    uint32_t st = 12345u;
    auto rnd = [&]() { // LCG -> floats in (-4, 4)
        st = st * 1664525u + 1013904223u;
        return ((float)(st >> 8) / 16777216.0f - 0.5f) * 8.0f;
    };
    for (int k = 0; k < 48; ++k) {
        const float ax = rnd(), ay = rnd(), az = rnd(), bx = rnd(), by = rnd(), bz = rnd(), t = rnd(); // (in this order)
        const vec3 a(ax, ay, az), b(bx, by, bz);
        const vec3 c = cross(a, b), n = b.normalized(), r = normalize(a - b), l = lerp(a, b, 0.5f + 0.1f * t), q = a * b + t * a - b / 3.0f;
        printf("%s[%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u]", k ? "," : "", bits(dot(a, b)), bits(a.length()),
               bits(a.length_squared()), bits(c.x), bits(c.y), bits(c.z), bits(n.x), bits(n.y), bits(n.z), bits(r.x), bits(r.y),
               bits(r.z), bits(l.x), bits(l.y), bits(l.z), bits(q.x), bits(q.y), bits(q.z));
    }
    printf("],\n");
    // common/matrix.cuh: mat3 * vec3, mat3 * mat3, transpose, determinant, inverse on 24 seeded matrices / vectors
    // (the tonemap multiplies the colour by two constant mat3s: render_utils.cuh:78-91)
    printf(" \"mat3_kat\": [");
    for (int k = 0; k < 24; ++k) {
        float a[9], b[9];
        for (float &x : a) x = rnd();
        for (float &x : b) x = rnd() * 0.25f;
        const vec3 v(rnd(), rnd(), rnd());
        const mat3 A(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]), B(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]);
        const vec3 av = A * v;
        const mat3 ab = A * B, at = A.transpose(), ai = A.inverse();
        printf("%s{\"a\":[", k ? "," : "");
        for (int i = 0; i < 9; ++i) printf("%s%u", i ? "," : "", bits(a[i]));
        printf("],\"b\":[");
        for (int i = 0; i < 9; ++i) printf("%s%u", i ? "," : "", bits(b[i]));
        printf("],\"v\":[%u,%u,%u],\"av\":[%u,%u,%u],\"det\":%u,\"ab\":[", bits(v.x), bits(v.y), bits(v.z), bits(av.x), bits(av.y),
               bits(av.z), bits(A.determinant()));
        for (int i = 0; i < 9; ++i) printf("%s%u", i ? "," : "", bits(ab.m[i / 3][i % 3]));
        printf("],\"at\":[");
        for (int i = 0; i < 9; ++i) printf("%s%u", i ? "," : "", bits(at.m[i / 3][i % 3]));
        printf("],\"ai\":[");
        for (int i = 0; i < 9; ++i) printf("%s%u", i ? "," : "", bits(ai.m[i / 3][i % 3]));
        printf("]}");
    }
    printf("],\n");
    // common/vec4.cuh on 24 seeded pairs; common/triangle.cuh on 32 seeded triangles, each with a ray aimed at a point of its
    // plane (inside for most, outside for some) so that both outcomes of intersect occur
    printf(" \"layout4\": {\"vec4\": %zu, \"vec4.w\": %zu, \"Triangle\": %zu, \"Triangle.v1\": %zu, \"Triangle.e1\": %zu, "
           "\"Triangle.e2\": %zu, \"Triangle.n\": %zu},\n",
           sizeof(vec4), offsetof(vec4, w), sizeof(Triangle), offsetof(Triangle, v1), offsetof(Triangle, e1), offsetof(Triangle, e2),
           offsetof(Triangle, n));
    printf(" \"vec4_kat\": [");
    for (int k = 0; k < 24; ++k) {
        const float ax = rnd(), ay = rnd(), az = rnd(), aw = rnd(), bx = rnd(), by = rnd(), bz = rnd(), bw = rnd(), t = rnd();
        const vec4 a(ax, ay, az, aw), b(vec3(bx, by, bz), bw);
        vec4 c = a;
        c += b;
        c *= t;
        c -= a;
        c /= bw;
        const vec4 q = (a + b) * t - b / aw + t * (-a), nn = normalize(b);
        const vec3 xyz = b.xyz();
        printf("%s[%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u]", k ? "," : "", bits(dot(a, b)), bits(length(a)), bits(c.x),
               bits(c.y), bits(c.z), bits(c.w), bits(q.x), bits(q.y), bits(q.z), bits(q.w), bits(nn.x), bits(nn.y), bits(nn.z), bits(nn.w),
               bits(xyz.x), bits(xyz.y), bits(xyz.z), bits(b[3]), bits(vec4(t)[2]));
    }
    printf("],\n \"triangle_kat\": [");
    for (int k = 0; k < 32; ++k) {
        float f[14];
        for (float &x : f) x = rnd(); // (in this order)
        const vec3 v0(f[0], f[1], f[2]), v1(f[3], f[4], f[5]), v2(f[6], f[7], f[8]), o(f[9], f[10], f[11] + 9.0f);
        const float wu = f[12] * 0.16f + 0.3f, wv = f[13] * 0.16f + 0.3f; // barycentrics in (-0.34, 0.94): some rays miss
        const Triangle T(v0, v1, v2);
        const vec3 target = v0 + wu * T.e1 + wv * T.e2, d = normalize(target - o);
        vec3 bmin, bmax;
        T.bounds(bmin, bmax);
        float t = 0.0f, u = 0.0f, v = 0.0f;
        const bool hit = T.intersect(Ray(o, d), t, u, v);
        const vec3 nrm = T.normal();
        printf("%s[%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%d,%u,%u,%u]", k ? "," : "", bits(T.e1.x), bits(T.e1.y),
               bits(T.e1.z), bits(T.e2.x), bits(T.e2.y), bits(T.e2.z), bits(T.n.x), bits(T.n.y), bits(T.n.z), bits(nrm.x), bits(nrm.y),
               bits(nrm.z), bits(T.area()), bits(bmin.x), bits(bmin.y), bits(bmin.z), bits(bmax.x), bits(bmax.y), bits(bmax.z), hit ? 1 : 0,
               hit ? bits(t) : 0u, hit ? bits(u) : 0u, hit ? bits(v) : 0u);
    }
    printf("]\n}\n");
    return 0;
}
