// ptrt/math.hpp -- host-side math PODs of the PTRT path tracer's Scene API.
//
// Mirrors the public types a caller of the reference's `Scene` touches
// (reference: src/common/vec3.cuh, src/common/mat4.cuh,
// src/pathtracer/scene/transform.cuh:14-146,148-306).  Host only: the device
// side lives behind the C ABI in include/ptrt.h.  Built with -ffp-contract=off
// so a*b+c here is two roundings, like the reference's MSVC host build.
#pragma once
#include <cmath>
#include <cstring>

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    vec3(float s) : x(s), y(s), z(s) {} // implicit, as in the reference (vec3.cuh:15)

    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }

    vec3 operator-() const { return {-x, -y, -z}; }
    vec3 operator+(const vec3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
    vec3 operator-(const vec3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
    vec3 operator*(const vec3 &o) const { return {x * o.x, y * o.y, z * o.z}; }
    vec3 operator*(float t) const { return {x * t, y * t, z * t}; }
    vec3 operator/(float t) const { return {x / t, y / t, z / t}; }
    vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3 &operator*=(float t) { x *= t; y *= t; z *= t; return *this; }

    float length_squared() const { return x * x + y * y + z * z; }
    float length() const { return std::sqrt(length_squared()); }
    vec3 normalized() const {
        float l = length();
        return l > 0 ? (*this / l) : vec3(0, 0, 0);
    }
};
inline vec3 operator*(float t, const vec3 &v) { return v * t; }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3 &a, const vec3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline vec3 normalize(const vec3 &v) { return v.normalized(); }
inline vec3 lerp(const vec3 &a, const vec3 &b, float t) { return (1.0f - t) * a + t * b; }
using point3 = vec3;

// src/common/vec4.cuh:13-122 -- the homogeneous point of the motion-vector stage (denoiser_kernels.cuh:54-56).  As in the
// reference, `/` and `/=` multiply by the rounded reciprocal (vec4.cuh:70-77, 101-103), unlike vec3's true division.
struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec3 xyz() const { return vec3(x, y, z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    vec4 operator-() const { return vec4(-x, -y, -z, -w); }
    vec4 &operator+=(const vec4 &v) { x += v.x; y += v.y; z += v.z; w += v.w; return *this; }
    vec4 &operator-=(const vec4 &v) { x -= v.x; y -= v.y; z -= v.z; w -= v.w; return *this; }
    vec4 &operator*=(float s) { x *= s; y *= s; z *= s; w *= s; return *this; }
    vec4 &operator/=(float s) { const float inv = 1.0f / s; x *= inv; y *= inv; z *= inv; w *= inv; return *this; }
};
inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4 &v, float s) { return vec4(v.x * s, v.y * s, v.z * s, v.w * s); }
inline vec4 operator*(float s, const vec4 &v) { return v * s; }
inline vec4 operator/(const vec4 &v, float s) { return v * (1.0f / s); }
inline float dot(const vec4 &a, const vec4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length(const vec4 &v) { return std::sqrt(dot(v, v)); }
inline vec4 normalize(const vec4 &v) { return v / length(v); }
inline float length(const vec3 &v) { return v.length(); } // vec3.cuh free function, used by Triangle::area

// src/common/ray.cuh:9-34
class Ray {
  public:
    point3 orig;
    vec3 dir;
    bool spec = false;
    Ray() {}
    Ray(const point3 &origin, const vec3 &direction) : orig(origin), dir(direction), spec(false) {}
    Ray(const point3 &origin, const vec3 &direction, bool is_specular) : orig(origin), dir(direction), spec(is_specular) {}
    point3 origin() const { return orig; }
    vec3 direction() const { return dir; }
    bool isSpecular() const { return spec; }
    point3 at(float t) const { return orig + t * dir; }
};

#ifndef PI
#define PI 3.14159265358979323846f
#endif
#ifndef TWO_PI
#define TWO_PI 6.28318530717958647692f
#endif

// 16 floats.  The reference documents column-major storage (mat4.cuh:15-21) but
// Transform3D and the ray transforms use m[0..3] as the first ROW
// (transform.cuh:297-299, intersection.cuh:258-263).  Both conventions are kept
// exactly where the reference uses them, including operator*'s `b.m[11]` slip in
// the m[3] term (mat4.cuh:289), which is harmless for affine matrices.
struct mat4 {
    float m[16];
    mat4() {
        std::memset(m, 0, sizeof m);
        m[0] = m[5] = m[10] = m[15] = 1.0f;
    }
    static mat4 identity() { return mat4(); }
    // right-handed view matrix, column-major (mat4.cuh:143-163)
    static mat4 lookAt(const vec3 &eye, const vec3 &center, const vec3 &up) {
        const vec3 f = normalize(center - eye);
        const vec3 s = normalize(cross(f, up));
        const vec3 u = cross(s, f);
        mat4 r;
        r.m[0] = s.x; r.m[4] = s.y; r.m[8] = s.z;
        r.m[1] = u.x; r.m[5] = u.y; r.m[9] = u.z;
        r.m[2] = -f.x; r.m[6] = -f.y; r.m[10] = -f.z;
        r.m[12] = -dot(s, eye);
        r.m[13] = -dot(u, eye);
        r.m[14] = dot(f, eye);
        return r;
    }
    // right-handed perspective projection (mat4.cuh:168-195)
    static mat4 perspective(float fov_y_radians, float aspect, float zNear, float zFar) {
        mat4 r;
        const float t = tanf(fov_y_radians / 2.0f);
        std::memset(r.m, 0, sizeof r.m);
        r.m[0] = 1.0f / (aspect * t);
        r.m[5] = 1.0f / (t);
        r.m[10] = -(zFar + zNear) / (zFar - zNear);
        r.m[11] = -1.0f;
        r.m[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
        return r;
    }
    mat4 transpose() const {
        mat4 r;
        for (int c = 0; c < 4; ++c)
            for (int k = 0; k < 4; ++k)
                r.m[c * 4 + k] = m[k * 4 + c];
        return r;
    }
    // analytic inverse by 2x2 sub-determinants (mat4.cuh:207-262); identity when |det| < 1e-10
    mat4 inverse() const {
        const float *a = m;
        float s2323 = a[10] * a[15] - a[11] * a[14], s1323 = a[9] * a[15] - a[11] * a[13];
        float s1223 = a[9] * a[14] - a[10] * a[13], s0323 = a[8] * a[15] - a[11] * a[12];
        float s0223 = a[8] * a[14] - a[10] * a[12], s0123 = a[8] * a[13] - a[9] * a[12];
        float s2313 = a[6] * a[15] - a[7] * a[14], s1313 = a[5] * a[15] - a[7] * a[13];
        float s1213 = a[5] * a[14] - a[6] * a[13], s0313 = a[4] * a[15] - a[7] * a[12];
        float s0213 = a[4] * a[14] - a[6] * a[12], s0113 = a[4] * a[13] - a[5] * a[12];
        float s2312 = a[6] * a[11] - a[7] * a[10], s1312 = a[5] * a[11] - a[7] * a[9];
        float s1212 = a[5] * a[10] - a[6] * a[9], s0312 = a[4] * a[11] - a[7] * a[8];
        float s0212 = a[4] * a[10] - a[6] * a[8], s0112 = a[4] * a[9] - a[5] * a[8];
        float det = a[0] * (a[5] * s2323 - a[6] * s1323 + a[7] * s1223) -
                    a[1] * (a[4] * s2323 - a[6] * s0323 + a[7] * s0223) +
                    a[2] * (a[4] * s1323 - a[5] * s0323 + a[7] * s0123) -
                    a[3] * (a[4] * s1223 - a[5] * s0223 + a[6] * s0123);
        if (std::fabs(det) < 1e-10f)
            return mat4();
        float id = 1.0f / det;
        mat4 r;
        r.m[0] = id * (a[5] * s2323 - a[6] * s1323 + a[7] * s1223);
        r.m[1] = id * -(a[1] * s2323 - a[2] * s1323 + a[3] * s1223);
        r.m[2] = id * (a[1] * s2313 - a[2] * s1313 + a[3] * s1213);
        r.m[3] = id * -(a[1] * s2312 - a[2] * s1312 + a[3] * s1212);
        r.m[4] = id * -(a[4] * s2323 - a[6] * s0323 + a[7] * s0223);
        r.m[5] = id * (a[0] * s2323 - a[2] * s0323 + a[3] * s0223);
        r.m[6] = id * -(a[0] * s2313 - a[2] * s0313 + a[3] * s0113);
        r.m[7] = id * (a[0] * s2312 - a[2] * s0312 + a[3] * s0112);
        r.m[8] = id * (a[4] * s1323 - a[5] * s0323 + a[7] * s0123);
        r.m[9] = id * -(a[0] * s1323 - a[1] * s0323 + a[3] * s0123);
        r.m[10] = id * (a[0] * s1313 - a[1] * s0313 + a[3] * s0113);
        r.m[11] = id * -(a[0] * s1312 - a[1] * s0312 + a[3] * s0112);
        r.m[12] = id * -(a[4] * s1223 - a[5] * s0223 + a[6] * s0123);
        r.m[13] = id * (a[0] * s1223 - a[1] * s0223 + a[2] * s0123);
        r.m[14] = id * -(a[0] * s1213 - a[1] * s0213 + a[2] * s0113);
        r.m[15] = id * (a[0] * s1212 - a[1] * s0212 + a[2] * s0112);
        return r;
    }
};
// column-major product as the reference writes it (mat4.cuh:279-323)
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) {
            const float *bc = &b.m[c * 4];
            float b1 = bc[1];
            if (c == 0 && k == 3)
                b1 = b.m[11]; // the reference's typo in r.m[3]
            r.m[c * 4 + k] = a.m[k] * bc[0] + a.m[4 + k] * b1 + a.m[8 + k] * bc[2] + a.m[12 + k] * bc[3];
        }
    return r;
}

struct AABB { // transform.cuh:14-146
    vec3 bmin, bmax;
    static AABB make_invalid() { return {vec3(1e30f), vec3(-1e30f)}; }
    vec3 extent() const { return bmax - bmin; }
    vec3 center() const { return (bmin + bmax) * 0.5f; }
    void expand(const vec3 &p) {
        bmin.x = fminf(bmin.x, p.x); bmin.y = fminf(bmin.y, p.y); bmin.z = fminf(bmin.z, p.z);
        bmax.x = fmaxf(bmax.x, p.x); bmax.y = fmaxf(bmax.y, p.y); bmax.z = fmaxf(bmax.z, p.z);
    }
    void expand(const AABB &b) {
        bmin.x = fminf(bmin.x, b.bmin.x); bmin.y = fminf(bmin.y, b.bmin.y); bmin.z = fminf(bmin.z, b.bmin.z);
        bmax.x = fmaxf(bmax.x, b.bmax.x); bmax.y = fmaxf(bmax.y, b.bmax.y); bmax.z = fmaxf(bmax.z, b.bmax.z);
    }
};

// Per-mesh instance transform (transform.cuh:148-417).
struct Transform3D {
    vec3 position{0.0f}, rotation{0.0f}, scale{1.0f};
    mat4 worldMatrix, inverseMatrix, normalMatrix;
    bool dirty = true;

    Transform3D() { updateMatrices(); }
    explicit Transform3D(vec3 p, vec3 r = vec3(0.0f), vec3 s = vec3(1.0f)) : position(p), rotation(r), scale(s) {
        updateMatrices();
    }
    void setPosition(const vec3 &p) { position = p; dirty = true; }
    void setRotation(const vec3 &r) { rotation = r; dirty = true; }
    void setScale(const vec3 &s) { scale = s; dirty = true; }
    void setScale(float s) { scale = vec3(s); dirty = true; }
    void translate(const vec3 &d) { position = position + d; dirty = true; }
    void rotate(const vec3 &d) { rotation = rotation + d; dirty = true; }

    // transform.cuh:260-306, reproduced operation by operation
    void updateMatrices() {
        if (!dirty)
            return;
        float cx = cosf(rotation.x), sx = sinf(rotation.x);
        float cy = cosf(rotation.y), sy = sinf(rotation.y);
        float cz = cosf(rotation.z), sz = sinf(rotation.z);
        mat4 rot;
        rot.m[0] = cy * cz;
        rot.m[1] = cz * sx * sy - cx * sz;
        rot.m[2] = cx * cz * sy + sx * sz;
        rot.m[3] = 0.0f;
        rot.m[4] = cy * sz;
        rot.m[5] = cx * cz + sx * sy * sz;
        rot.m[6] = cx * sy * sz - cz * sx;
        rot.m[7] = 0.0f;
        rot.m[8] = -sy;
        rot.m[9] = cy * sx;
        rot.m[10] = cx * cy;
        rot.m[11] = 0.0f;
        rot.m[12] = rot.m[13] = rot.m[14] = 0.0f;
        rot.m[15] = 1.0f;
        worldMatrix = mat4::identity();
        worldMatrix.m[0] = scale.x;
        worldMatrix.m[5] = scale.y;
        worldMatrix.m[10] = scale.z;
        worldMatrix = rot * worldMatrix;
        worldMatrix.m[3] = position.x;
        worldMatrix.m[7] = position.y;
        worldMatrix.m[11] = position.z;
        inverseMatrix = worldMatrix.inverse();
        normalMatrix = inverseMatrix.transpose();
        dirty = false;
    }
    vec3 transformPoint(const vec3 &p) const {
        const float *w = worldMatrix.m;
        return {w[0] * p.x + w[1] * p.y + w[2] * p.z + w[3], w[4] * p.x + w[5] * p.y + w[6] * p.z + w[7],
                w[8] * p.x + w[9] * p.y + w[10] * p.z + w[11]};
    }
    // transform.cuh:399-416: world AABB = bounds of the 8 transformed corners
    AABB transformAABB(const AABB &b) const {
        AABB out = AABB::make_invalid();
        for (int i = 0; i < 8; ++i)
            out.expand(transformPoint(vec3((i & 1) ? b.bmax.x : b.bmin.x, (i & 2) ? b.bmax.y : b.bmin.y,
                                           (i & 4) ? b.bmax.z : b.bmin.z)));
        return out;
    }
};
