/*
 * oracle/detmath.h -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked
 * into the product).
 *
 * Deterministic fp32 elementary functions built only from IEEE-754 operations
 * that are correctly rounded on both x86-64 and gfx950: + - * / sqrt, fused
 * multiply-add, round-to-nearest-even, and integer bit manipulation.  The HIP
 * kernels carry their own copy of the same algorithms
 * (ptrt-game-engine_amd/csrc/det_math.hip.h); tests/test_detmath_gpu.py checks
 * the two bit for bit, tests/test_detmath.py checks this file against glibc.
 *
 * They stand in for the CUDA libm calls on the reference's path
 * (sinf/cosf: math/sampling.cuh:115,155,201 and rendering/pbr_utils.cuh:120;
 *  expf/logf: rendering/pbr_utils.cuh:149-161 via path_logic.cuh:826-828;
 *  powf: scene/scene.cuh:2031-2039).  CUDA's own results for these differ from
 * glibc's and from ROCm's by ulps, so no choice of libm reproduces the CUDA
 * binary; choosing one arithmetic for oracle AND kernel is what makes the
 * radiance comparison exact instead of statistical.  Accuracy: <= 2 ulp on the
 * ranges the path uses, pow <= 5 ulp (measured in tests/test_detmath.py).
 */
#ifndef PTRT_ORACLE_DETMATH_H
#define PTRT_ORACLE_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float dm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

static inline uint32_t dm_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float dm_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* CUDA fmaxf/fminf semantics (a NaN operand is ignored); the sign of a zero
 * result is "second operand on ties", identical in the kernel copy. */
static inline float dm_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float dm_min(float a, float b) { return (a < b || b != b) ? a : b; }

/* sin and cos of x (radians), |x| up to a few thousand. */
static inline void dm_sincos(float x, float *s_out, float *c_out) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    *s_out = sinf(x);
    *c_out = cosf(x);
    return;
#endif
    const float kf = rintf(x * 0x1.45f306p-1f); /* 2/pi */
    const int k = (int)kf;
    float r = dm_fma(-kf, 0x1.921fb6p+0f, x); /* pi/2 in three pieces */
    r = dm_fma(-kf, -0x1.777a5cp-25f, r);
    r = dm_fma(-kf, -0x1.ee59dap-50f, r);
    const float r2 = r * r;
    float sp = dm_fma(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = dm_fma(sp, r2, -1.6666654611e-1f);
    const float sn = dm_fma(sp * r2, r, r);
    float cp = dm_fma(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = dm_fma(cp, r2, 4.166664568298827e-2f);
    const float cs = dm_fma(cp, r2 * r2, dm_fma(-0.5f, r2, 1.0f));
    float s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = s;
    *c_out = c;
}
static inline float dm_sin(float x) {
    float s, c;
    dm_sincos(x, &s, &c);
    return s;
}
static inline float dm_cos(float x) {
    float s, c;
    dm_sincos(x, &s, &c);
    return c;
}

/* e^x */
static inline float dm_exp(float x) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    return expf(x);
#endif
    if (x != x)
        return x;
    if (x > 88.72283f)
        return dm_float(0x7f800000u);
    if (x < -104.0f)
        return 0.0f;
    const float kf = rintf(x * 0x1.715476p+0f); /* log2(e) */
    float r = dm_fma(-kf, 0x1.62e4p-1f, x);     /* ln2 hi (few bits: k*hi exact) */
    r = dm_fma(-kf, 0x1.7f7d1cp-20f, r);        /* ln2 lo */
    float p = 1.9875691500e-4f;
    p = dm_fma(p, r, 1.3981999507e-3f);
    p = dm_fma(p, r, 8.3334519073e-3f);
    p = dm_fma(p, r, 4.1665795894e-2f);
    p = dm_fma(p, r, 1.6666665459e-1f);
    p = dm_fma(p, r, 5.0000001201e-1f);
    p = dm_fma(p, r * r, r) + 1.0f;
    const int k = (int)kf;         /* -150 .. 128 */
    const int k1 = k >> 1;         /* arithmetic shift */
    const int k2 = k - k1;
    const float s1 = dm_float((uint32_t)(k1 + 127) << 23);
    const float s2 = dm_float((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

/* natural log; x <= 0 follows IEEE (log 0 = -inf, log negative = NaN) */
static inline float dm_log(float x) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    return logf(x);
#endif
    if (x != x)
        return x;
    if (x < 0.0f)
        return dm_float(0x7fc00000u);
    if (x == 0.0f)
        return dm_float(0xff800000u);
    uint32_t u = dm_bits(x);
    if (u == 0x7f800000u)
        return x;
    int e = 0;
    if (u < 0x00800000u) { /* denormal: scale by 2^23 */
        x = x * 8388608.0f;
        u = dm_bits(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;                               /* x = m * 2^e, m in [0.5,1) */
    float m = dm_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = dm_fma(y, m, -1.1514610310e-1f);
    y = dm_fma(y, m, 1.1676998740e-1f);
    y = dm_fma(y, m, -1.2420140846e-1f);
    y = dm_fma(y, m, 1.4249322787e-1f);
    y = dm_fma(y, m, -1.6668057665e-1f);
    y = dm_fma(y, m, 2.0000714765e-1f);
    y = dm_fma(y, m, -2.4999993993e-1f);
    y = dm_fma(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = dm_fma(fe, -2.12194440e-4f, y);
    y = dm_fma(-0.5f, z, y);
    float r = m + y;
    r = dm_fma(fe, 0.693359375f, r);
    return r;
}

/* x^y for x > 0 (the only use on the path: the sRGB OETF, x in (0.003,1]) */
static inline float dm_pow(float x, float y) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    return powf(x, y);
#endif
    return dm_exp(y * dm_log(x));
}

/* atan for x >= 0 (Cephes atanf: two range reductions, degree-4 polynomial in x^2) */
static inline float dm_atan_pos(float x) {
    float y = 0.0f;
    if (x > 0x1.3504f4p+1f) { /* tan(3 pi/8) */
        y = 0x1.921fb6p+0f;   /* pi/2 */
        x = -(1.0f / x);
    } else if (x > 0x1.a8279ap-2f) { /* tan(pi/8) */
        y = 0x1.921fb6p-1f;          /* pi/4 */
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    float p = dm_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = dm_fma(p, z, 1.99777106478e-1f);
    p = dm_fma(p, z, -3.33329491539e-1f);
    return y + dm_fma(p * z, x, x);
}
/* atan2f(y, x) in [-pi, pi] (sampleSky's phi, rendering/render_utils.cuh:126); NaN in -> NaN out */
static inline float dm_atan2(float y, float x) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    return atan2f(y, x);
#endif
    if (x != x || y != y)
        return x + y;
    const float PI_F = 0x1.921fb6p+1f, PIO2_F = 0x1.921fb6p+0f;
    if (x == 0.0f) {
        if (y == 0.0f)
            return (dm_bits(x) >> 31) ? ((dm_bits(y) >> 31) ? -PI_F : PI_F) : y;
        return y < 0.0f ? -PIO2_F : PIO2_F;
    }
    if (y == 0.0f)
        return x < 0.0f ? ((dm_bits(y) >> 31) ? -PI_F : PI_F) : y;
    const float ax = fabsf(x), ay = fabsf(y);
    float a; /* atan(|y|/|x|) in [0, pi/2] */
    if (ax == INFINITY)
        a = (ay == INFINITY) ? 0x1.921fb6p-1f : 0.0f;
    else if (ay == INFINITY)
        a = PIO2_F;
    else
        a = dm_atan_pos(ay / ax);
    if (x < 0.0f)
        a = PI_F - a;
    return y < 0.0f ? -a : a;
}
/* acosf(x), |x| <= 1 (sampleSky's theta, render_utils.cuh:128; the argument is clamped there).
 * Cephes asinf: x or sqrt((1-|x|)/2) into a degree-4 polynomial in its square. */
static inline float dm_asin_core(float a, float z) { /* a + a*z*P(z) */
    float p = dm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = dm_fma(p, z, 4.5470025998e-2f);
    p = dm_fma(p, z, 7.4953002686e-2f);
    p = dm_fma(p, z, 1.6666752422e-1f);
    return dm_fma(p * z, a, a);
}
static inline float dm_acos(float x) {
#ifdef PTRT_ORACLE_LIBM /* (tools/fmad_sensitivity.py --variant libm: the platform libm in place of the deterministic version) */
    return acosf(x);
#endif
    if (!(fabsf(x) <= 1.0f))
        return NAN;
    const float PI_F = 0x1.921fb6p+1f, PIO2_F = 0x1.921fb6p+0f;
    if (x > 0.5f) {
        const float z = 0.5f * (1.0f - x);
        return 2.0f * dm_asin_core(sqrtf(z), z);
    }
    if (x < -0.5f) {
        const float z = 0.5f * (1.0f + x);
        return PI_F - 2.0f * dm_asin_core(sqrtf(z), z);
    }
    return PIO2_F - dm_asin_core(x, x * x);
}

#endif
