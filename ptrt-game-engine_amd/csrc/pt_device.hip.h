// pt_device.hip.h -- gfx950 device function library of the path tracer: vector
// helpers, deterministic elementary functions, the XORWOW generator and the
// shading model (BSDF evaluation / pdf / sampling, next-event estimation terms,
// tonemap).
//
// Arithmetic contract (DESIGN.md "Numerics"): the translation unit is built with
// -ffp-contract=off; dot/cross/length^2 use explicit fused multiply-adds in one
// fixed association; everything else is a rounded multiply followed by a
// rounded add; division and sqrt are IEEE correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); sin/cos/exp/log/pow are the
// polynomial versions below, built from those operations only.  The CPU oracle
// is written to the same contract, so every buffer the kernels produce can be
// compared bit for bit.
//
// Reference semantics implemented here (file:line of Mark-Rindler/PTRT-game-engine):
//   rendering/path_logic.cuh:44-52,73-122,157-250,305-393,490-780
//   math/pdf.cuh:26-30,73-220   math/sampling.cuh:15-43,73-91,105-120,141-164,187-208
//   rendering/pbr_utils.cuh:16-161   rendering/render_utils.cuh:21-45,77-95,115-125
//   rendering/taa.cuh:19-61   scene/scene.cuh:2004-2047   cuRAND XORWOW (published algorithm)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace pt {

// ------------------------------------------------------------------ vectors
struct f3 {
    float x, y, z;
};
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 mk3(float s) { return f3{s, s, s}; }
PT_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PT_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV f3 operator*(f3 a, float t) { return f3{a.x * t, a.y * t, a.z * t}; }
PT_DEV f3 operator*(float t, f3 a) { return f3{a.x * t, a.y * t, a.z * t}; }
PT_DEV f3 operator/(f3 a, float t); // (three quotients by one divisor: div3, below)
PT_DEV f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_DEV f3 operator+(f3 a, float t) { return f3{a.x + t, a.y + t, a.z + t}; }
PT_DEV f3 operator-(f3 a, float t) { return f3{a.x - t, a.y - t, a.z - t}; }

PT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ------------------------------------------------ correctly rounded 1/y and x/y, cheaper
// hipcc expands an IEEE fp32 division into v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup
// (11 VALU).  The scale/fixup steps only act on extreme exponents; for operands whose exponents are
// in [2^-60, 2^60) the Newton core alone gives the same correctly rounded result:
//   1/y : r0 = v_rcp_f32(y) (<= 1 ulp); e = fma(-y,r0,1); r = fma(e,r0,r0)             (3 VALU)
//   x/y : q0 = x*r; q1 = fma(fma(-y,q0,x), r, q0); q = fma(fma(-y,q1,x), r, q1)          (+5 VALU)
// tests/test_misc_gpu.py checks rcp_ieee against `1.0f/y` for EVERY fp32 bit pattern and div_ieee
// against `x/y` on 2^32 random pairs; anything outside the guarded range takes the compiler's
// division.  The oracle divides with the CPU's IEEE divider.
PT_DEV bool mid_exponent(float x) { return (((__float_as_uint(x) >> 23) & 0xffu) - 67u) < 120u; }
PT_DEV float rcp_ieee(float y) {
    // (the Newton core runs unconditionally and the rare wave with an extreme exponent patches its lanes afterwards:
    // written as `if (extreme) return 1/y;` first, the compiler laid out three branches in front of the common path)
    const float r0 = __builtin_amdgcn_rcpf(y);
    const float e = fma_(-y, r0, 1.0f);
    float r = fma_(e, r0, r0);
    const bool mid = mid_exponent(y);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!mid) != 0ull, 0))
        r = mid ? r : 1.0f / y;
    return r;
}
// v / t for a vector: the reference divides component by component (vec3.cuh:54-56), so three IEEE divisions share their
// divisor -- and hipcc's expansion of each (above) shares nothing, because v_div_scale looks at both operands.  With the
// correctly rounded reciprocal r = RN(1/t) of rcp_ieee's Newton core, ONE residual step per component is enough
// (Markstein): q0 = RN(a r); rho = a - q0 t (exact in an FMA); q = RN(q0 + rho r) = RN(a / t).  That is a statement about
// significands only -- every step is an IEEE operation, so it scales with the operands' exponents as long as nothing
// leaves the normal range -- and tests/test_div3_gpu.py checks it on the GPU for EVERY pair of significands (2^46
// pairs, tools/div3_exhaustive.py; a sample of divisors in the suite) against the compiler's division.  Exponents:
// t in [2^-40, 2^40] and each component 0 or in [2^-60, 2^60] keep q0, rho and q normal and rho exact; anything else (and a
// NaN or infinite component, which the magnitude test rejects or v_div_fixup resolves exactly as in hipcc's sequence) takes
// the compiler's division, wave-uniformly.  v_div_fixup also gives a zero numerator its sign (q0 + rho r would turn -0
// into +0).  16 VALU + 9 of guards instead of 36: Cornell 1080p 2.13 -> 2.01 ms (without the guards 1.96); the showcase
// kernel, bound by latency rather than by VALU issue, pays 1 % for the guards' branches (4.10 -> 4.14).
PT_DEV float div3_core(float a, float t, float r) {
    const float q0 = a * r;
    return __builtin_amdgcn_div_fixupf(fma_(fma_(-t, q0, a), r, q0), t, a);
}
PT_DEV bool div3_in_range(f3 a, float t) {
    // components: zero (either sign) or |a| >= 2^-60 -- as integers, (bits << 1) - 2 sends +-0 to the top and orders the rest
    const uint32_t lo = min(min(__float_as_uint(a.x) * 2u - 2u, __float_as_uint(a.y) * 2u - 2u), __float_as_uint(a.z) * 2u - 2u);
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const float at = __builtin_fabsf(t);
    return lo >= (0x21800000u * 2u - 2u) && hi <= 0x1p60f && at >= 0x1p-40f && at <= 0x1p40f;
}
// (out of line: inlined, the three divisions of the fallback were ~40 instructions at each of ~30 call sites -- 8 KB more
// code per kernel and most of the gain gone)
__device__ __attribute__((noinline)) f3 div3_slow(f3 a, float t) { return f3{a.x / t, a.y / t, a.z / t}; }
PT_DEV f3 div3(f3 a, float t) {
    const float r0 = __builtin_amdgcn_rcpf(t);
    const float r = fma_(fma_(-t, r0, 1.0f), r0, r0);
    f3 q = f3{div3_core(a.x, t, r), div3_core(a.y, t, r), div3_core(a.z, t, r)};
    const bool ok = div3_in_range(a, t);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
        const f3 s = div3_slow(a, t);
        q.x = ok ? q.x : s.x;
        q.y = ok ? q.y : s.y;
        q.z = ok ? q.z : s.z;
    }
    return q;
}
PT_DEV f3 operator/(f3 a, float t) { return div3(a, t); }
// Correctly rounded sqrt(x), cheaper: hipcc's expansion is v_sqrt_f32, two FMA tests of the neighbouring floats, a pre- and
// post-scaling for tiny arguments and a class test (16 VALU + the transcendental).  For a mid-range x the reciprocal square
// root's first-order correction already rounds correctly: r = v_rsq_f32(x); g = x r; h = r / 2; s = g + (x - g g) h, the
// residual exact in an FMA.  tests/test_misc_gpu.py compares sqrt_ieee with the compiler's sqrtf for EVERY fp32 bit pattern
// (PT_SQRT_VARIANT 2 adds a coupled Newton step to g and h first); zero, subnormal, huge, negative, infinite and NaN
// arguments take the compiler's sequence, wave-uniformly.
#ifndef PT_SQRT_VARIANT
#define PT_SQRT_VARIANT 1
#endif
PT_DEV float sqrt_core(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
#if PT_SQRT_VARIANT == 2
    const float e = fma_(-h, g, 0.5f);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
#endif
    return fma_(fma_(-g, g, x), h, g);
}
PT_DEV float sqrt_ieee(float x) {
    float s = sqrt_core(x);
    const bool mid = (__float_as_uint(x) - 0x21800000u) < (0x5d800000u - 0x21800000u); // positive, 2^-60 <= x < 2^60
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!mid) != 0ull, 0))
        s = mid ? s : __builtin_sqrtf(x);
    return s;
}
PT_DEV float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PT_DEV f3 cross(f3 a, f3 b) {
    return f3{fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
PT_DEV float length(f3 v) { return sqrt_ieee(dot(v, v)); }
// v / |v|: sqrt_ieee and div3 under ONE guard.  A squared length in [2^-60, 2^60) puts the length in [2^-30, 2^30] -- inside
// div3's divisor range -- and bounds every component by it; what is left to check is that no component is a tiny nonzero
// number.  Everything else (zero vectors included: the reference returns 0 for them) takes the compiler's sqrtf and divisions.
__device__ __attribute__((noinline)) f3 normalize_slow(f3 v) {
    const float len = __builtin_sqrtf(dot(v, v));
    return (len > 0) ? f3{v.x / len, v.y / len, v.z / len} : mk3(0.0f);
}
PT_DEV f3 normalize(f3 v) {
    const float d = dot(v, v);
    const float len = sqrt_core(d);
    const float r0 = __builtin_amdgcn_rcpf(len);
    const float r = fma_(fma_(-len, r0, 1.0f), r0, r0);
    f3 q = f3{div3_core(v.x, len, r), div3_core(v.y, len, r), div3_core(v.z, len, r)};
    const uint32_t lo = min(min(__float_as_uint(v.x) * 2u - 2u, __float_as_uint(v.y) * 2u - 2u), __float_as_uint(v.z) * 2u - 2u);
    const bool ok = (__float_as_uint(d) - 0x21800000u) < (0x5d800000u - 0x21800000u) && lo >= (0x21800000u * 2u - 2u);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0ull, 0)) {
        const f3 s = normalize_slow(v);
        q.x = ok ? q.x : s.x;
        q.y = ok ? q.y : s.y;
        q.z = ok ? q.z : s.z;
    }
    return q;
}
// NaN-ignoring max/min with "second operand on ties" (matches the oracle's dm_max/dm_min)
PT_DEV float max_(float a, float b) { return (a > b || b != b) ? a : b; }
PT_DEV float min_(float a, float b) { return (a < b || b != b) ? a : b; }
PT_DEV float clamp01(float x) { return min_(max_(x, 0.0f), 1.0f); }
PT_DEV float clampf(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
PT_DEV f3 clampv(f3 v, float lo, float hi) {
    return f3{min_(max_(v.x, lo), hi), min_(max_(v.y, lo), hi), min_(max_(v.z, lo), hi)};
}
PT_DEV f3 lerp(f3 a, f3 b, float t) { return (1.0f - t) * a + t * b; }
PT_DEV f3 reflectVec(f3 I, f3 N) { return I - 2.0f * dot(I, N) * N; }

constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;

// ------------------------------------------------------ deterministic math
PT_DEV void det_sincos(float x, float &s_out, float &c_out) {
    const float kf = __builtin_rintf(x * 0x1.45f306p-1f);
    const int k = (int)kf;
    float r = fma_(-kf, 0x1.921fb6p+0f, x);
    r = fma_(-kf, -0x1.777a5cp-25f, r);
    r = fma_(-kf, -0x1.ee59dap-50f, r);
    const float r2 = r * r;
    float sp = fma_(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fma_(sp, r2, -1.6666654611e-1f);
    const float sn = fma_(sp * r2, r, r);
    float cp = fma_(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fma_(cp, r2, 4.166664568298827e-2f);
    const float cs = fma_(cp, r2 * r2, fma_(-0.5f, r2, 1.0f));
    const bool swap = (k & 1) != 0;
    float s = swap ? cs : sn;
    float c = swap ? sn : cs;
    if (k & 2)
        s = -s;
    if ((k + 1) & 2)
        c = -c;
    s_out = s;
    c_out = c;
}
PT_DEV float det_cos(float x) {
    float s, c;
    det_sincos(x, s, c);
    return c;
}
PT_DEV float det_sin(float x) {
    float s, c;
    det_sincos(x, s, c);
    return s;
}
PT_DEV float det_exp(float x) {
    // (the three special cases are selects AFTER the core, not branches around it: for an argument outside the range the core
    // computes garbage that is then replaced -- same bits as the early returns of oracle/detmath.h.  Measured against the
    // branches: temporal_kernel 114 -> 103 us, an a-trous pass 88 -> 90 us, path tracing unchanged)
    const float x_in = x;
    x = (x >= -104.0f && x <= 88.72283f) ? x : 0.0f;
    const float kf = __builtin_rintf(x * 0x1.715476p+0f);
    float r = fma_(-kf, 0x1.62e4p-1f, x);
    r = fma_(-kf, 0x1.7f7d1cp-20f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    p = fma_(p, r * r, r) + 1.0f;
    const int k = (int)kf;
    const int k1 = k >> 1;
    const int k2 = k - k1;
    const float s1 = __uint_as_float((uint32_t)(k1 + 127) << 23);
    const float s2 = __uint_as_float((uint32_t)(k2 + 127) << 23);
    float e = (p * s1) * s2;
    e = (x_in < -104.0f) ? 0.0f : e;
    e = (x_in > 88.72283f) ? __uint_as_float(0x7f800000u) : e;
    return (x_in != x_in) ? x_in : e;
}
PT_DEV float det_log(float x) {
    if (x != x)
        return x;
    if (x < 0.0f)
        return __uint_as_float(0x7fc00000u);
    if (x == 0.0f)
        return __uint_as_float(0xff800000u);
    uint32_t u = __float_as_uint(x);
    if (u == 0x7f800000u)
        return x;
    int e = 0;
    if (u < 0x00800000u) {
        x = x * 8388608.0f;
        u = __float_as_uint(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fma_(y, m, -1.1514610310e-1f);
    y = fma_(y, m, 1.1676998740e-1f);
    y = fma_(y, m, -1.2420140846e-1f);
    y = fma_(y, m, 1.4249322787e-1f);
    y = fma_(y, m, -1.6668057665e-1f);
    y = fma_(y, m, 2.0000714765e-1f);
    y = fma_(y, m, -2.4999993993e-1f);
    y = fma_(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fma_(fe, -2.12194440e-4f, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(fe, 0.693359375f, r);
    return r;
}
PT_DEV float det_pow(float x, float y) { return det_exp(y * det_log(x)); }

// atan for x >= 0 (Cephes atanf), atan2f and acosf for sampleSky's equirect lookup
// (rendering/render_utils.cuh:126-128); same algorithms as oracle/detmath.h
PT_DEV float det_atan_pos(float x) {
    float y = 0.0f;
    if (x > 0x1.3504f4p+1f) { // tan(3 pi/8)
        y = 0x1.921fb6p+0f;
        x = -(1.0f / x);
    } else if (x > 0x1.a8279ap-2f) { // tan(pi/8)
        y = 0x1.921fb6p-1f;
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fma_(p, z, 1.99777106478e-1f);
    p = fma_(p, z, -3.33329491539e-1f);
    return y + fma_(p * z, x, x);
}
PT_DEV float det_atan2(float y, float x) {
    if (x != x || y != y)
        return x + y;
    const float PI_F = 0x1.921fb6p+1f, PIO2_F = 0x1.921fb6p+0f;
    if (x == 0.0f) {
        if (y == 0.0f)
            return (__float_as_uint(x) >> 31) ? ((__float_as_uint(y) >> 31) ? -PI_F : PI_F) : y;
        return y < 0.0f ? -PIO2_F : PIO2_F;
    }
    if (y == 0.0f)
        return x < 0.0f ? ((__float_as_uint(y) >> 31) ? -PI_F : PI_F) : y;
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float INF = __builtin_inff();
    float a;
    if (ax == INF)
        a = (ay == INF) ? 0x1.921fb6p-1f : 0.0f;
    else if (ay == INF)
        a = PIO2_F;
    else
        a = det_atan_pos(ay / ax);
    if (x < 0.0f)
        a = PI_F - a;
    return y < 0.0f ? -a : a;
}
PT_DEV float det_asin_core(float a, float z) {
    float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fma_(p, z, 4.5470025998e-2f);
    p = fma_(p, z, 7.4953002686e-2f);
    p = fma_(p, z, 1.6666752422e-1f);
    return fma_(p * z, a, a);
}
PT_DEV float det_acos(float x) {
    if (!(__builtin_fabsf(x) <= 1.0f))
        return __builtin_nanf("");
    const float PI_F = 0x1.921fb6p+1f, PIO2_F = 0x1.921fb6p+0f;
    if (x > 0.5f) {
        const float z = 0.5f * (1.0f - x);
        return 2.0f * det_asin_core(sqrt_ieee(z), z);
    }
    if (x < -0.5f) {
        const float z = 0.5f * (1.0f + x);
        return PI_F - 2.0f * det_asin_core(sqrt_ieee(z), z);
    }
    return PIO2_F - det_asin_core(x, x * x);
}

// tex2D<float4>(envMap, u, v) of the texture object Scene::loadHDRI creates (scene.cuh:1007-1013:
// normalised coordinates, wrap in u / clamp in v, linear filter), restated from the CUDA C
// Programming Guide's texture-fetching appendix: xB = frac(u)*W - 0.5, i = floor(xB), the weight
// frac(xB) kept with 8 fractional bits; texel indices outside the map follow the address mode.
PT_DEV f3 env_texel(const float4 *env, int w, int h, int i, int j) {
    i %= w;
    if (i < 0)
        i += w;
    j = j < 0 ? 0 : (j > h - 1 ? h - 1 : j);
    const float4 t = env[(size_t)j * w + i];
    return f3{t.x, t.y, t.z};
}
PT_DEV f3 tex2d_env(const float4 *env, int w, int h, float u, float v) {
    const float uw = u - __builtin_floorf(u);
    const float vmax = 1.0f - 1.0f / (float)h;
    const float vc = v < 0.0f ? 0.0f : (v >= 1.0f ? vmax : v);
    const float xB = uw * (float)w - 0.5f, yB = vc * (float)h - 0.5f;
    const float fi = __builtin_floorf(xB), fj = __builtin_floorf(yB);
    const float a = __builtin_rintf((xB - fi) * 256.0f) * (1.0f / 256.0f),
                b = __builtin_rintf((yB - fj) * 256.0f) * (1.0f / 256.0f);
    const int i = (int)fi, j = (int)fj;
    const f3 t00 = env_texel(env, w, h, i, j), t10 = env_texel(env, w, h, i + 1, j), t01 = env_texel(env, w, h, i, j + 1),
             t11 = env_texel(env, w, h, i + 1, j + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return ((t00 * w00 + t10 * w10) + t01 * w01) + t11 * w11;
}

// ------------------------------------------------------------------ XORWOW
struct Rng {
    uint32_t d, v0, v1, v2, v3, v4;
};
PT_DEV uint32_t rng_next(Rng &s) {
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
// curand_uniform: (0,1]
PT_DEV float rng_uniform(Rng &s) { return (float)rng_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

// ---------------------------------------------------------------- materials
// One record = 6 float4 (see ptrt_capi.hip: pack_material)
struct Material {
    f3 albedo, specular, emission, sheenTint;
    float metallic, roughness, transmission, ior, transmissionRoughness, clearcoat, clearcoatRoughness, iridescence,
        iridescenceThickness, sheen;
};
// A load through a pointer KNOWN to point into LDS: a `cond ? lds : global` choice of base otherwise compiles to flat
// loads of a selected generic pointer.
typedef float vec4f_t __attribute__((ext_vector_type(4)));
typedef float vec2f_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const vec4f_t lds_vec4f;
typedef __attribute__((address_space(3))) const vec2f_t lds_vec2f;
PT_DEV float4 lds_ld4(const float4 *p, int i) {
    const vec4f_t v = ((lds_vec4f *)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef __attribute__((address_space(3))) const float lds_f32;
PT_DEV float lds_ld1(const float *p, int i) { return ((lds_f32 *)p)[i]; }
PT_DEV float2 lds_ld2(const float2 *p, int i) {
    const vec2f_t v = ((lds_vec2f *)p)[i];
    return make_float2(v.x, v.y);
}
template <bool LDS> PT_DEV float4 ld4(const float4 *p, int i) { return LDS ? lds_ld4(p, i) : p[i]; }

// (STRIDE 3: records cut down to what the simple-material kernels read -- albedo, metallic, specular, roughness, emission,
// transmission; the rest reads as zero and is not looked at)
template <bool LDS = false, int STRIDE = 6> PT_DEV Material load_material(const float4 *__restrict__ recs, int id) {
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 a = ld4<LDS>(recs, id * STRIDE + 0), b = ld4<LDS>(recs, id * STRIDE + 1), c = ld4<LDS>(recs, id * STRIDE + 2),
                 d = STRIDE > 3 ? ld4<LDS>(recs, id * STRIDE + 3) : z, e = STRIDE > 4 ? ld4<LDS>(recs, id * STRIDE + 4) : z,
                 f = STRIDE > 5 ? ld4<LDS>(recs, id * STRIDE + 5) : z;
    Material m;
    m.albedo = mk3(a.x, a.y, a.z);
    m.metallic = a.w;
    m.specular = mk3(b.x, b.y, b.z);
    m.roughness = b.w;
    m.emission = mk3(c.x, c.y, c.z);
    m.transmission = c.w;
    m.sheenTint = mk3(d.x, d.y, d.z);
    m.ior = d.w;
    m.transmissionRoughness = e.x;
    m.clearcoat = e.y;
    m.clearcoatRoughness = e.z;
    m.iridescence = e.w;
    m.iridescenceThickness = f.x;
    m.sheen = f.y;
    return m;
}

struct Surface { // what the shading functions need of HitInfo
    f3 point, normal;
    float t;
    bool front_face;
};

// ------------------------------------------------------------- PBR helpers
PT_DEV f3 fresnelSchlick(float cosTheta, f3 F0) {
    cosTheta = clamp01(cosTheta);
    const float f = 1.0f - cosTheta;
    const float f2 = f * f;
    const float f5 = f2 * f2 * f;
    return F0 + (mk3(1.0f) - F0) * f5;
}
PT_DEV float distributionGGX(f3 N, f3 H, float roughness) {
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float NdotH = max_(dot(N, H), 0.0f);
    const float NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = PI_F * denom * denom;
    return a2 / max_(denom, 1e-6f);
}
PT_DEV float geometrySchlickGGX(float NdotV, float roughness) {
    const float r = (roughness + 1.0f);
    const float k = (r * r) * 0.125f;
    return NdotV / (NdotV * (1.0f - k) + k + 1e-6f);
}
PT_DEV float geometrySmith(f3 N, f3 V, f3 L, float roughness) {
    const float NdotV = max_(dot(N, V), 0.0f);
    const float NdotL = max_(dot(N, L), 0.0f);
    const float ggx2 = geometrySchlickGGX(NdotV, roughness);
    const float ggx1 = geometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}
PT_DEV float geometrySmithTransmission(f3 N, f3 V, f3 L, float roughness) {
    const float NdotV = max_(dot(N, V), 0.0f);
    const float NdotL = __builtin_fabsf(dot(N, L));
    const float ggx2 = geometrySchlickGGX(NdotV, roughness);
    const float ggx1 = geometrySchlickGGX(NdotL, roughness);
    return ggx1 * ggx2;
}
PT_DEV f3 calculateIridescence(float thickness, float cosTheta, float filmIOR, float baseIOR) {
    cosTheta = clamp01(cosTheta);
    const float sinTheta = sqrt_ieee(1.0f - cosTheta * cosTheta);
    const float sinThetaFilm = sinTheta / filmIOR;
    if (sinThetaFilm * sinThetaFilm > 1.0f)
        return mk3(1.0f);
    const float cosThetaFilm = sqrt_ieee(1.0f - sinThetaFilm * sinThetaFilm);
    const float OPD = 2.0f * filmIOR * thickness * cosThetaFilm;
    float Ra = (1.0f - filmIOR) / (1.0f + filmIOR);
    Ra *= Ra;
    float Rb = (filmIOR - baseIOR) / (filmIOR + baseIOR);
    Rb *= Rb;
    const float sqrtR1R2 = sqrt_ieee(Ra * Rb);
    float R_max = (sqrt_ieee(Ra) + sqrt_ieee(Rb));
    R_max *= R_max;
    const float inv_R_max = 1.0f / (R_max + 1e-6f);
    const float d0 = TWO_PI_F * OPD * (1.0f / 650.0f);
    const float d1 = TWO_PI_F * OPD * (1.0f / 550.0f);
    const float d2 = TWO_PI_F * OPD * (1.0f / 450.0f);
    const float r0 = Ra + Rb + 2.0f * sqrtR1R2 * det_cos(d0);
    const float r1 = Ra + Rb + 2.0f * sqrtR1R2 * det_cos(d1);
    const float r2 = Ra + Rb + 2.0f * sqrtR1R2 * det_cos(d2);
    return mk3(clamp01(r0 * inv_R_max), clamp01(r1 * inv_R_max), clamp01(r2 * inv_R_max));
}
PT_DEV float schlick_dielectric(float cosTheta, float ior_i, float ior_t) {
    cosTheta = clamp01(cosTheta);
    float r0 = (ior_i - ior_t) / (ior_i + ior_t);
    r0 = r0 * r0;
    const float f = 1.0f - cosTheta;
    const float f2 = f * f;
    const float f5 = f2 * f2 * f;
    return r0 + (1.0f - r0) * f5;
}
PT_DEV f3 beerLambert(f3 ac, float dist) {
    const f3 c = mk3(max_(ac.x, 0.0f), max_(ac.y, 0.0f), max_(ac.z, 0.0f));
    return mk3(det_exp(-c.x * dist), det_exp(-c.y * dist), det_exp(-c.z * dist));
}
PT_DEV float attenuate(float distance, float range) {
    const float att = range / (range + distance);
    return att * att;
}
PT_DEV f3 clamp_vector_soft(f3 v, float max_lum) {
    const float lum = 0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z;
    if (lum > max_lum && lum > 0.0f) {
        const float scale = max_lum / lum;
        return v * scale;
    }
    return v;
}

// F0 shared by evaluateBSDF / material_pdf / material_scatter
template <bool FULL> PT_DEV f3 base_F0(const Material &mat, float metal, float NdotV) {
    f3 F0 = lerp(mat.specular, mat.albedo, metal);
    if (FULL) {
        const float irid = clamp01(mat.iridescence);
        if (irid > 0.0f) {
            const f3 ic = calculateIridescence(mat.iridescenceThickness, NdotV, 1.3f, mat.ior);
            F0 = lerp(F0, ic, irid);
        }
    }
    return F0;
}

// ---------------------------------------------------------------- sampling
PT_DEV void orthoBasis(f3 N, f3 &T, f3 &B) {
    const float len2 = dot(N, N);
    if (len2 < 1e-20f) {
        T = mk3(1.0f, 0.0f, 0.0f);
        B = mk3(0.0f, 1.0f, 0.0f);
        return;
    }
    const f3 Nn = N * (1.0f / sqrt_ieee(len2));
    const float s = __builtin_copysignf(1.0f, Nn.z);
    const float a = -1.0f / (s + Nn.z);
    const float b = Nn.x * Nn.y * a;
    T = mk3(1.0f + s * Nn.x * Nn.x * a, s * b, -s * Nn.x);
    B = cross(Nn, T);
}
PT_DEV f3 to_world(f3 sample, f3 N) {
    f3 T, B;
    orthoBasis(N, T, B);
    return sample.x * T + sample.y * B + sample.z * N;
}
PT_DEV f3 sample_cone_direction(Rng &rng, f3 cone_dir, float cos_theta_max) {
    const float u1 = rng_uniform(rng);
    const float u2 = rng_uniform(rng);
    const float cos_theta = 1.0f - u1 * (1.0f - cos_theta_max);
    const float sin_theta = sqrt_ieee(max_(0.0f, 1.0f - cos_theta * cos_theta));
    const float phi = TWO_PI_F * u2;
    float sp, cp;
    det_sincos(phi, sp, cp);
    return to_world(mk3(sin_theta * cp, sin_theta * sp, cos_theta), cone_dir);
}
PT_DEV f3 sample_cosine_hemisphere(Rng &rng) {
    const float u1 = rng_uniform(rng);
    const float u2 = rng_uniform(rng);
    const float r = sqrt_ieee(u1);
    const float phi = TWO_PI_F * u2;
    float sp, cp;
    det_sincos(phi, sp, cp);
    return mk3(r * cp, r * sp, sqrt_ieee(max_(0.0f, 1.0f - u1)));
}
PT_DEV f3 importance_sample_ggx(Rng &rng, f3 N, float roughness) {
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float u1 = rng_uniform(rng);
    float u2 = rng_uniform(rng);
    u2 = min_(u2, 0.9999999f);
    const float phi = TWO_PI_F * u1;
    const float cosTheta = sqrt_ieee((1.0f - u2) / (1.0f + (a2 - 1.0f) * u2));
    const float sinTheta = sqrt_ieee(max_(0.0f, 1.0f - cosTheta * cosTheta));
    float sp, cp;
    det_sincos(phi, sp, cp);
    return to_world(mk3(sinTheta * cp, sinTheta * sp, cosTheta), N);
}

// --------------------------------------------------------------- BSDF eval
template <bool FULL> PT_DEV f3 evaluateBSDF(const Surface &hit, const Material &mat, f3 L, f3 V) {
    const f3 N = hit.normal;
    const float NdotV = max_(dot(N, V), 0.0f);
    if (NdotV <= 0.0f)
        return mk3(0.0f);
    const float metal = clamp01(mat.metallic);
    const float rough = max_(mat.roughness, 0.02f);
    const f3 F0_base = base_F0<FULL>(mat, metal, NdotV);
    if (FULL) {
        const float trans = clamp01(mat.transmission);
        if (trans > 0.0f && metal < 0.1f) {
            const float ior = mat.ior;
            const float transRough = max_(mat.transmissionRoughness, rough);
            const float eta = hit.front_face ? (1.0f / ior) : ior;
            const float NdotL = dot(N, L);
            if (NdotL > 0.0f) {
                const f3 H = normalize(L + V);
                const float VdotH = max_(dot(V, H), 0.0f);
                const float D = distributionGGX(N, H, rough);
                const float G = geometrySmith(N, V, L, rough);
                const f3 F = fresnelSchlick(VdotH, F0_base);
                const f3 spec = (D * G * F) / (4.0f * NdotV * NdotL + 1e-6f);
                return spec * NdotL;
            }
            f3 H = normalize(-(V * eta + L));
            if (dot(N, H) < 0.0f)
                H = -H;
            const float VdotH = max_(dot(V, H), 0.0f);
            const float LdotH = __builtin_fabsf(dot(L, H));
            const float NdotL_abs = __builtin_fabsf(NdotL);
            const float k = 1.0f - eta * eta * (1.0f - VdotH * VdotH);
            if (k < 0.0f)
                return mk3(0.0f);
            const float D = distributionGGX(N, H, transRough);
            const float G = geometrySmithTransmission(N, V, L, transRough);
            const f3 F = mk3(1.0f) - fresnelSchlick(VdotH, F0_base);
            const float numerator = (eta * eta * (1.0f - metal) * G * D * VdotH * LdotH);
            const float pw = eta * VdotH + LdotH;
            const float denominator = NdotV * NdotL_abs * (pw * pw);
            const f3 btdf = (mat.albedo * F * numerator) / (denominator + 1e-6f);
            return btdf * NdotL_abs;
        }
    }
    const float NdotL = max_(dot(N, L), 0.0f);
    if (NdotL <= 0.0f)
        return mk3(0.0f);
    const f3 H = normalize(L + V);
    const float VdotH = max_(dot(V, H), 0.0f);
    const float D = distributionGGX(N, H, rough);
    const float G = geometrySmith(N, V, L, rough);
    const f3 F = fresnelSchlick(VdotH, F0_base);
    const f3 specular = (D * G * F) / (4.0f * NdotV * NdotL + 0.001f);
    const f3 kD = (mk3(1.0f) - F) * (1.0f - metal);
    const f3 diffuse = kD * mat.albedo / PI_F;
    return (diffuse + specular) * NdotL;
}

// --------------------------------------------------------------------- pdfs
PT_DEV float mis_weight(float pdf1, float pdf2) {
    const float a = pdf1 * pdf1;
    const float b = pdf2 * pdf2;
    return a / (a + b + 1e-10f);
}
PT_DEV float pdf_ggx_reflect(f3 N, f3 V, f3 L, float roughness) {
    const float NdotV = max_(dot(N, V), 0.0f);
    if (NdotV == 0.0f)
        return 0.0f;
    const f3 H = normalize(V + L);
    const float NdotH = max_(dot(N, H), 0.0f);
    const float VdotH = max_(dot(V, H), 0.0f);
    const float D = distributionGGX(N, H, roughness);
    const float pdf_H = D * NdotH;
    // (the constant through an opaque move: as a literal the vectoriser pairs `4 VdotH` with `D NdotH` into one v_pk_mul_f32 and
    // the loop-invariant {-, 4.0} operand pair became a register pair parked in scratch across the render loop)
    float four = 4.0f;
    asm volatile("" : "+v"(four));
    return pdf_H / (four * VdotH + 1e-6f);
}
PT_DEV float pdf_ggx_refract(f3 N, f3 V, f3 L, float roughness, float eta) {
    const float NdotV = max_(dot(N, V), 0.0f);
    const float NdotL = dot(N, L);
    if (NdotV <= 0.0f || NdotL >= 0.0f)
        return 0.0f;
    f3 H = normalize(-(V * eta + L));
    if (dot(N, H) < 0.0f)
        H = -H;
    const float VdotH = max_(dot(V, H), 0.0f);
    const float LdotH = __builtin_fabsf(dot(L, H));
    const float NdotH = max_(dot(N, H), 0.0f);
    const float D = distributionGGX(N, H, roughness);
    const float pdf_H = D * NdotH;
    const float pw = eta * VdotH + LdotH;
    const float dwh_dwo = (eta * eta * LdotH) / (pw * pw);
    return pdf_H * __builtin_fabsf(dwh_dwo);
}
template <bool FULL> PT_DEV float material_pdf(const Surface &hit, const Material &mat, f3 V, f3 L) {
    const f3 N = hit.normal;
    const float NdotV = max_(dot(N, V), 0.0f);
    const float NdotL = max_(dot(N, L), 0.0f);
    if (NdotV == 0.0f)
        return 0.0f;
    const float metal = clamp01(mat.metallic);
    const float rough = max_(mat.roughness, 0.02f);
    const f3 F0_base = base_F0<FULL>(mat, metal, NdotV);
    const f3 F_base = fresnelSchlick(NdotV, F0_base);
    float total_pdf = 0.0f;
    float prob_base = 1.0f;
    if (FULL) {
        const float clearcoat = clamp01(mat.clearcoat);
        if (clearcoat > 0.0f) {
            const float ccr = max_(mat.clearcoatRoughness, 0.001f);
            const f3 F_coat = fresnelSchlick(NdotV, mk3(0.04f));
            const float F_coat_avg = (F_coat.x + F_coat.y + F_coat.z) * (1.0f / 3.0f);
            const float P_coat = clamp01(F_coat_avg * clearcoat);
            if (NdotL > 0.0f)
                total_pdf += P_coat * pdf_ggx_reflect(N, V, L, ccr);
            prob_base = (1.0f - P_coat);
        }
        const float trans = clamp01(mat.transmission);
        if (trans > 0.0f && metal < 0.1f) {
            const float ior = mat.ior;
            const float transRough = max_(mat.transmissionRoughness, rough);
            const float ior_ratio = hit.front_face ? (1.0f / ior) : ior;
            const float reflect_prob = schlick_dielectric(NdotV, 1.0f, ior_ratio);
            if (NdotL > 0.0f) {
                const float pdf_reflect = pdf_ggx_reflect(N, V, L, rough);
                total_pdf += prob_base * reflect_prob * pdf_reflect;
                const f3 H = normalize(V + L);
                const float VdotH = max_(dot(V, H), 0.0f);
                const float k = 1.0f - ior_ratio * ior_ratio * (1.0f - VdotH * VdotH);
                if (k < 0.0f) {
                    const float p2 = pdf_ggx_reflect(N, V, L, transRough);
                    total_pdf += prob_base * (1.0f - reflect_prob) * p2;
                }
            } else {
                const float pdf_refract = pdf_ggx_refract(N, V, L, transRough, ior_ratio);
                total_pdf += prob_base * (1.0f - reflect_prob) * pdf_refract;
            }
            return total_pdf;
        }
    }
    if (NdotL > 0.0f) {
        const float max_fresnel = max_(F_base.x, max_(F_base.y, F_base.z));
        const float specular_prob = (metal > 0.0f) ? 1.0f : max_fresnel;
        const float pdf_spec = pdf_ggx_reflect(N, V, L, rough);
        const float pdf_diffuse = NdotL * (1.0f / PI_F);
        total_pdf += prob_base * (specular_prob * pdf_spec + (1.0f - specular_prob) * pdf_diffuse);
    }
    return total_pdf;
}

// ------------------------------------------------------------ BSDF sampling
// returns false when the path ends (no lobe left to sample)
template <bool FULL>
// `draws_only`: the vertex is the path's last (the depth limit follows): nothing of the scattered ray will be looked at, only
// the generator has to move on exactly as it would -- the same uniforms drawn, the same answer, none of the arithmetic.
PT_DEV bool material_scatter(const Surface &hit, const Material &mat, f3 ray_dir, Rng &rng, f3 &scattered_dir,
                             f3 &attenuation, bool &is_specular_bounce, bool draws_only = false) {
    const f3 V = -ray_dir;
    const f3 N = hit.normal;
    const float NdotV = max_(dot(N, V), 0.0f);
    const float metal = clamp01(mat.metallic);
    const float rough = max_(mat.roughness, 0.02f);
    const f3 albedo = mat.albedo;
    const f3 F0_base = base_F0<FULL>(mat, metal, NdotV);
    const f3 F_base_for_NdotV = fresnelSchlick(NdotV, F0_base);
    float clearcoat = 0.0f;
    float P_coat = 0.0f;
    float prob_base = 1.0f;
    float clearcoatRough = 0.0f;
    f3 F0_coat = mk3(0.0f);
    if (FULL) {
        clearcoat = clamp01(mat.clearcoat);
        if (clearcoat > 0.0f) {
            clearcoatRough = max_(mat.clearcoatRoughness, 0.001f);
            F0_coat = mk3(0.04f);
            const f3 F_coat = fresnelSchlick(NdotV, F0_coat);
            const float F_coat_avg = (F_coat.x + F_coat.y + F_coat.z) * (1.0f / 3.0f);
            P_coat = clamp01(F_coat_avg * clearcoat);
            prob_base = (1.0f - P_coat);
        }
        const float trans = clamp01(mat.transmission);
        if (trans > 0.0f && metal < 0.1f) {
            const float ior = mat.ior;
            const float transRough = max_(mat.transmissionRoughness, rough);
            const float eta = hit.front_face ? (1.0f / ior) : ior;
            const float ior_incident = hit.front_face ? 1.0f : ior;
            const float ior_transmitted = hit.front_face ? ior : 1.0f;
            const float reflect_prob = schlick_dielectric(NdotV, ior_incident, ior_transmitted);
            const float refract_prob = 1.0f - reflect_prob;
            const float P_trans_reflect = prob_base * reflect_prob;
            const float P_trans_refract = prob_base * refract_prob;
            const float u = rng_uniform(rng);
            if (draws_only) { // (importance_sample_ggx: two uniforms)
                (void)rng_uniform(rng);
                (void)rng_uniform(rng);
                return true;
            }
            // one GGX half-vector draw whichever lobe is chosen; only the roughness differs
            const bool pick_coat = u < P_coat;
            const bool pick_refl = !pick_coat && (u < P_coat + P_trans_reflect);
            const bool is_refraction = !pick_coat && !pick_refl;
            const float sample_roughness = pick_coat ? clearcoatRough : (pick_refl ? rough : transRough);
            f3 H = importance_sample_ggx(rng, N, sample_roughness);
            is_specular_bounce = (sample_roughness < 0.1f);
            if (!is_refraction) {
                scattered_dir = reflectVec(-V, H);
            } else {
                float VdotH_tir = dot(V, H);
                if (VdotH_tir < 0.0f)
                    H = -H;
                VdotH_tir = __builtin_fabsf(dot(V, H));
                const float k = 1.0f - eta * eta * (1.0f - VdotH_tir * VdotH_tir);
                if (k < 0.0f) {
                    scattered_dir = reflectVec(-V, H);
                    is_specular_bounce = true;
                } else {
                    const float cos_t = sqrt_ieee(k);
                    scattered_dir = normalize(eta * (-V) + (eta * VdotH_tir - cos_t) * H);
                }
            }
            const float NdotL = dot(N, scattered_dir);
            f3 f_total = mk3(0.0f);
            float pdf_total = 0.0f;
            f3 F_coat_atten;
            if (is_refraction) {
                const f3 Hb = normalize(eta * V + scattered_dir);
                F_coat_atten = fresnelSchlick(max_(dot(V, Hb), 0.0f), F0_coat);
            } else {
                const f3 Hb = normalize(V + scattered_dir);
                F_coat_atten = fresnelSchlick(max_(dot(V, Hb), 0.0f), F0_coat);
            }
            const f3 base_attenuation = mk3(1.0f) - clearcoat * F_coat_atten;
            if (P_coat > 0.0f && NdotL > 0.0f) {
                const f3 Hc = normalize(V + scattered_dir);
                const float NdotHc = max_(dot(N, Hc), 0.0f);
                const float VdotHc = max_(dot(V, Hc), 0.0f);
                const float Dc = distributionGGX(N, Hc, clearcoatRough);
                const float Gc = geometrySmith(N, V, scattered_dir, clearcoatRough);
                const f3 Fc = fresnelSchlick(VdotHc, F0_coat);
                const float pdf_L_coat = (Dc * NdotHc) / (4.0f * VdotHc + 1e-6f);
                pdf_total += P_coat * pdf_L_coat;
                const f3 brdf_coat = (Dc * Gc * Fc) / (4.0f * NdotV * NdotL + 1e-6f);
                f_total = f_total + clearcoat * brdf_coat * NdotL;
            }
            if (P_trans_reflect > 0.0f && NdotL > 0.0f) {
                const f3 Hr = normalize(V + scattered_dir);
                const float NdotHr = max_(dot(N, Hr), 0.0f);
                const float VdotHr = max_(dot(V, Hr), 0.0f);
                const float Dr = distributionGGX(N, Hr, rough);
                const float Gr = geometrySmith(N, V, scattered_dir, rough);
                const f3 Fr = fresnelSchlick(VdotHr, F0_base);
                const float pdf_L_refl = (Dr * NdotHr) / (4.0f * VdotHr + 1e-6f);
                pdf_total += P_trans_reflect * pdf_L_refl;
                const f3 brdf_refl = (Dr * Gr * Fr) / (4.0f * NdotV * NdotL + 1e-6f);
                f_total = f_total + brdf_refl * NdotL * base_attenuation;
            }
            if (P_trans_refract > 0.0f && NdotL < 0.0f) {
                f3 Ht = normalize(-(V * eta + scattered_dir));
                if (dot(N, Ht) < 0.0f)
                    Ht = -Ht;
                const float VdotHt = max_(dot(V, Ht), 0.0f);
                const float LdotHt = __builtin_fabsf(dot(scattered_dir, Ht));
                const float NdotHt = max_(dot(N, Ht), 0.0f);
                const float NdotL_abs = __builtin_fabsf(NdotL);
                const float k = 1.0f - eta * eta * (1.0f - VdotHt * VdotHt);
                if (k >= 0.0f) {
                    const float Dt = distributionGGX(N, Ht, transRough);
                    const float Gt = geometrySmithTransmission(N, V, scattered_dir, transRough);
                    const float pw = eta * VdotHt + LdotHt;
                    const float dwh_dwo = (eta * eta * LdotHt) / (pw * pw);
                    const float pdf_L_refr = (Dt * NdotHt * __builtin_fabsf(dwh_dwo));
                    pdf_total += P_trans_refract * pdf_L_refr;
                    const f3 Ft = mk3(1.0f) - fresnelSchlick(VdotHt, F0_base);
                    const float numerator = (eta * eta * (1.0f - metal) * Gt * Dt * VdotHt * LdotHt);
                    const float denominator = NdotV * NdotL_abs * (pw * pw);
                    const f3 btdf = (albedo * Ft * numerator) / (denominator + 1e-6f);
                    f_total = f_total + btdf * NdotL_abs * base_attenuation;
                }
            }
            if (is_refraction && NdotL > 0.0f) {
                const f3 Hr = normalize(V + scattered_dir);
                const float NdotHr = max_(dot(N, Hr), 0.0f);
                const float VdotHr = max_(dot(V, Hr), 0.0f);
                const float Dr = distributionGGX(N, Hr, transRough);
                const float Gr = geometrySmith(N, V, scattered_dir, transRough);
                const float pdf_L_refl = (Dr * NdotHr) / (4.0f * VdotHr + 1e-6f);
                pdf_total += P_trans_refract * pdf_L_refl;
                const f3 brdf_refl = (Dr * Gr * mk3(1.0f)) / (4.0f * NdotV * NdotL + 1e-6f);
                f_total = f_total + brdf_refl * NdotL * base_attenuation;
            }
            const float out_pdf = max_(pdf_total, 1e-6f);
            attenuation = f_total / out_pdf;
            return true;
        }
    }

    const float max_fresnel = max_(F_base_for_NdotV.x, max_(F_base_for_NdotV.y, F_base_for_NdotV.z));
    const float specular_prob = (metal > 0.0f) ? 1.0f : max_fresnel;
    const float P_opaque_spec = prob_base * specular_prob;
    const float P_opaque_diff = prob_base * (1.0f - specular_prob);
    const float u = rng_uniform(rng);
    const bool pick_coat = FULL && (u < P_coat);
    const bool pick_spec = !pick_coat && (u < P_coat + P_opaque_spec);
    const bool ggx = pick_coat || pick_spec;
    if (!ggx && !(P_opaque_diff > 1e-6f))
        return false;
    {
        // importance_sample_ggx (sampling.cuh:187-208) for the lanes that picked a specular lobe and sample_cosine_hemisphere
        // (:141-164) for the others in ONE pass: both draw two uniforms, take the sine and cosine of 2 pi times one of them,
        // two square roots, and turn a local vector into N's frame -- a wave that holds both kinds (most do: the lobe is a
        // per-lane draw) used to run the two functions one after the other.  Every lane performs exactly the operations of
        // its own function, on its own operands.
        const float r = pick_coat ? clearcoatRough : rough;
        const float a = r * r, a2 = a * a;
        const float u1 = rng_uniform(rng);
        const float u2r = rng_uniform(rng);
        if (draws_only)
            return true;
        const float u2 = ggx ? min_(u2r, 0.9999999f) : u2r;
        float sp, cp;
        det_sincos(TWO_PI_F * (ggx ? u1 : u2), sp, cp);
        // ggx: s1 = cos(theta) = sqrt((1 - u2) / (1 + (a2 - 1) u2)), s2 = sin(theta) = sqrt(max(0, 1 - s1 s1));
        // cosine: s1 = radius = sqrt(u1), s2 = z = sqrt(max(0, 1 - u1))
        const float s1 = sqrt_ieee(ggx ? (1.0f - u2) / (1.0f + (a2 - 1.0f) * u2) : u1);
        const float s2 = sqrt_ieee(max_(0.0f, 1.0f - (ggx ? s1 * s1 : u1)));
        const float rad = ggx ? s2 : s1, up = ggx ? s1 : s2;
        const f3 w = to_world(mk3(rad * cp, rad * sp, up), N);
        const f3 refl = reflectVec(-V, w);
        scattered_dir.x = ggx ? refl.x : w.x;
        scattered_dir.y = ggx ? refl.y : w.y;
        scattered_dir.z = ggx ? refl.z : w.z;
        is_specular_bounce = ggx && (r < 0.1f);
    }
    scattered_dir = normalize(scattered_dir);
    const float NdotL = max_(dot(N, scattered_dir), 0.0f);
    f3 f_total = mk3(0.0f);
    float pdf_total = 0.0f;
    if (FULL && P_coat > 0.0f) {
        const f3 Hc = normalize(V + scattered_dir);
        const float NdotHc = max_(dot(N, Hc), 0.0f);
        const float VdotHc = max_(dot(V, Hc), 0.0f);
        const float Dc = distributionGGX(N, Hc, clearcoatRough);
        const float Gc = geometrySmith(N, V, scattered_dir, clearcoatRough);
        const f3 Fc = fresnelSchlick(VdotHc, F0_coat);
        const float pdf_L_coat = (Dc * NdotHc) / (4.0f * VdotHc + 1e-6f);
        pdf_total += P_coat * pdf_L_coat;
        const f3 brdf_coat = (Dc * Gc * Fc) / (4.0f * NdotV * NdotL + 1e-6f);
        f_total = f_total + clearcoat * brdf_coat * NdotL;
    }
    const f3 H_for_base = normalize(V + scattered_dir);
    const float VdotH_for_base = max_(dot(V, H_for_base), 0.0f);
    const f3 F_coat_atten = fresnelSchlick(VdotH_for_base, F0_coat);
    const f3 base_attenuation = mk3(1.0f) - clearcoat * F_coat_atten;
    const float NdotH_spec = max_(dot(N, H_for_base), 0.0f);
    const float D_spec = distributionGGX(N, H_for_base, rough);
    const float G_spec = geometrySmith(N, V, scattered_dir, rough);
    const f3 F_spec = fresnelSchlick(VdotH_for_base, F0_base);
    const float pdf_L_spec = (D_spec * NdotH_spec) / (4.0f * VdotH_for_base + 1e-6f);
    pdf_total += P_opaque_spec * pdf_L_spec;
    const f3 brdf_spec = (D_spec * G_spec * F_spec) / (4.0f * NdotV * NdotL + 1e-6f);
    f_total = f_total + brdf_spec * NdotL * base_attenuation;
    if (P_opaque_diff > 1e-6f) {
        const float pdf_L_diff = NdotL / PI_F;
        pdf_total += P_opaque_diff * pdf_L_diff;
        const f3 kD = (mk3(1.0f) - F_base_for_NdotV) * (1.0f - metal);
        f3 f_diff = (kD * albedo / PI_F) * NdotL;
        if (FULL) {
            const float sheen = clamp01(mat.sheen);
            if (sheen > 0.0f) {
                const float FH = 1.0f - max_(dot(V, H_for_base), 0.0f);
                const float FH5 = FH * FH * FH * FH * FH;
                const f3 Csheen = lerp(mk3(1.0f), mat.sheenTint, 0.5f);
                f_diff = f_diff + sheen * Csheen * FH5 * NdotL;
            }
        }
        f_total = f_total + f_diff * base_attenuation;
    }
    attenuation = f_total / max_(pdf_total, 1e-6f);
    return true;
}

// ------------------------------------------------------------------ jitter
// Halton(2,3) table of 16 as written in the reference (entry 15 repeats x = 0.0625), minus the 0.5 the caller subtracts
PT_DEV float2 taa_table_entry(int idx) {
    constexpr float HX[16] = {0.500000f, 0.250000f, 0.750000f, 0.125000f, 0.625000f, 0.375000f, 0.875000f, 0.062500f,
                              0.562500f, 0.312500f, 0.812500f, 0.187500f, 0.687500f, 0.437500f, 0.937500f, 0.062500f};
    constexpr float HY[16] = {0.333333f, 0.666667f, 0.111111f, 0.444444f, 0.777778f, 0.222222f, 0.555556f, 0.888889f,
                              0.037037f, 0.370370f, 0.703704f, 0.148148f, 0.481481f, 0.814815f, 0.259259f, 0.592593f};
    return make_float2(HX[idx] - 0.5f, HY[idx] - 0.5f);
}
PT_DEV void taa_jitter(int frame_index, float &jx, float &jy) {
    const float2 e = taa_table_entry(frame_index % 16);
    jx = e.x;
    jy = e.y;
}
// the frame's toroidal shift of the pixel's blue-noise value `val` (= table[(y & 63) * 64 + (x & 63)])
PT_DEV void blue_noise_shift(float2 val, int frame, float &ou, float &ov) {
    uint32_t hash = (uint32_t)frame * 0x9e3779b9u;
    hash ^= (hash >> 15);
    hash *= 0x85ebca6bu;
    hash ^= (hash >> 13);
    hash *= 0xc2b2ae35u;
    hash ^= (hash >> 16);
    const float shift_x = (float)(hash & 0xFFFFFF) * (1.0f / 16777216.0f);
    hash *= 0x85ebca6bu;
    const float shift_y = (float)(hash & 0xFFFFFF) * (1.0f / 16777216.0f);
    float u = val.x + shift_x;
    float v = val.y + shift_y;
    if (u >= 1.0f)
        u -= 1.0f;
    if (v >= 1.0f)
        v -= 1.0f;
    ou = u;
    ov = v;
}
PT_DEV void blue_noise_jitter(const float2 *__restrict__ table, int x, int y, int frame, float &ou, float &ov) {
    blue_noise_shift(table[(y & 63) * 64 + (x & 63)], frame, ou, ov);
}

// ----------------------------------------------------------------- tonemap
PT_DEV f3 aces_tonemap(f3 c) {
    f3 a = mk3(0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z,
               0.02840f * c.x + 0.13383f * c.y + 0.83777f * c.z);
    const f3 n = a * (a + 0.0245786f) - 0.000090537f;
    const f3 d = a * (0.983729f * a + 0.4329510f) + 0.238081f;
    a = clampv(n / d, 0.0f, 1.0f);
    a = mk3(1.60475f * a.x + -0.53108f * a.y + -0.07367f * a.z, -0.10208f * a.x + 1.10813f * a.y + -0.00605f * a.z,
            -0.00327f * a.x + -0.07276f * a.y + 1.07602f * a.z);
    return clampv(a, 0.0f, 1.0f);
}
PT_DEV float srgb_oetf(float c) {
    return (c <= 0.0031308f) ? 12.92f * c : 1.055f * det_pow(c, 1.0f / 2.4f) - 0.055f;
}
PT_DEV void tonemap_pixel(f3 hdr, unsigned char &r, unsigned char &g, unsigned char &b) {
    f3 color = hdr / 1.0f; // total_samples == 1 (scene.cuh:1204)
    color = aces_tonemap(color);
    color.x = srgb_oetf(color.x);
    color.y = srgb_oetf(color.y);
    color.z = srgb_oetf(color.z);
    const f3 rgb = clampv(color, 0.f, 1.f) * 255.99f;
    r = (unsigned char)rgb.x;
    g = (unsigned char)rgb.y;
    b = (unsigned char)rgb.z;
}


// Per-function probe for tests (ptrt_debug_shade; SURVEY 8(c) golden item 7: scalar known answers for a11-a13).
//   op 0: item = {material id, N(3), V(3), L(3), front_face}  -> {evaluateBSDF(3), material_pdf}
//   op 1: item = {material id, N(3), ray_dir(3), front_face, generator state (6 words)} ->
//                {scattered_dir(3), attenuation(3), flags (1 ok | 2 specular), generator state after (6 words)}
template <bool FULL> __global__ void shade_probe_kernel(const float4 *__restrict__ materials, int op, const float *__restrict__ in,
                                                        int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (op == 0) {
        const float *p = in + (size_t)i * 11;
        const Material mat = load_material(materials, (int)p[0]);
        Surface h;
        h.point = mk3(0.0f);
        h.normal = mk3(p[1], p[2], p[3]);
        h.t = 1.0f;
        h.front_face = p[10] != 0.0f;
        const f3 V = mk3(p[4], p[5], p[6]), L = mk3(p[7], p[8], p[9]);
        const f3 f = evaluateBSDF<FULL>(h, mat, L, V);
        float *o = out + (size_t)i * 4;
        o[0] = f.x;
        o[1] = f.y;
        o[2] = f.z;
        o[3] = material_pdf<FULL>(h, mat, V, L);
    } else {
        const float *p = in + (size_t)i * 14;
        const Material mat = load_material(materials, (int)p[0]);
        Surface h;
        h.point = mk3(0.0f);
        h.normal = mk3(p[1], p[2], p[3]);
        h.t = 1.0f;
        h.front_face = p[7] != 0.0f;
        Rng rng;
        rng.d = __float_as_uint(p[8]);
        rng.v0 = __float_as_uint(p[9]);
        rng.v1 = __float_as_uint(p[10]);
        rng.v2 = __float_as_uint(p[11]);
        rng.v3 = __float_as_uint(p[12]);
        rng.v4 = __float_as_uint(p[13]);
        f3 dir = mk3(0.0f), att = mk3(0.0f);
        bool spec = false;
        const bool ok = material_scatter<FULL>(h, mat, mk3(p[4], p[5], p[6]), rng, dir, att, spec);
        float *o = out + (size_t)i * 13;
        o[0] = dir.x;
        o[1] = dir.y;
        o[2] = dir.z;
        o[3] = att.x;
        o[4] = att.y;
        o[5] = att.z;
        o[6] = (float)((ok ? 1 : 0) | (spec ? 2 : 0));
        o[7] = __uint_as_float(rng.d);
        o[8] = __uint_as_float(rng.v0);
        o[9] = __uint_as_float(rng.v1);
        o[10] = __uint_as_float(rng.v2);
        o[11] = __uint_as_float(rng.v3);
        o[12] = __uint_as_float(rng.v4);
    }
}

} // namespace pt
