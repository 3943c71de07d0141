/*
 * oracle/kfo_math.h -- scalar number-format helpers of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (koifish_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Restates (does not include) the reference's number handling:
 *   - bf16 = upper 16 bits of an IEEE f32 (floatX = floatGama = __nv_bfloat16, src/g_float.hpp:246-262);
 *     every bf16 store of the oracle is round-to-nearest-even (the CUDA path's seeded stochastic
 *     stores, packedN.cuh:62-72, are a documented deviation: SURVEY.md fact 5).
 *   - f8e5m2 = top byte of an IEEE half (src/g_float.hpp:355-383 T2Float<f8e5>, :433-443 Float2T<f8e5>).
 *   - kfo_expf: a fixed, portable fp32 exp built from IEEE mul/add/fma only, so that the HIP kernels
 *     (koifish_amd/csrc/kf_math.h holds an independent statement of the same recipe) and the oracle
 *     agree bit for bit; the reference calls CUDA expf (Activation.cu:91, operator.cuh:267), a 2-ulp
 *     routine of its own.  tests/test_oracle_math.py pins kfo_expf to libm expf within 2 ulp.
 */
#ifndef KFO_MATH_H
#define KFO_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float kfo_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* round-to-nearest-even; NaN stays NaN (quiet) */
static inline uint16_t kfo_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float kfo_round_bf16(float f) { return kfo_bf16_to_f32(kfo_f32_to_bf16(f)); }

/* IEEE binary16 -> f32 (portable; gcc 11 has no _Float16 on x86) */
static inline float kfo_half_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) {
            u = s;
        } else { /* subnormal: normalise */
            int sh = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                sh++;
            }
            m &= 0x3ffu;
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 127 - 15) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* f32 -> IEEE binary16, round-to-nearest-even (what _cvtss_sh(x,0) does, GST_float.cpp:62) */
static inline uint16_t kfo_f32_to_half(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);      /* NaN */
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u);     /* overflow -> inf (>= 65536) */
    if (a < 0x33000000u) return (uint16_t)s;                  /* < 2^-25 -> 0 */
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t half_m = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
    uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
    /* half_m carries the implicit bit when normal; adding handles mantissa overflow into the exponent */
    uint32_t r = (e < -14) ? half_m : ((he << 10) + (half_m - 0x400u));
    if (r >= 0x7c00u) r = 0x7c00u;
    return (uint16_t)(s | r);
}

/* src/g_float.hpp:355-383: byte is the HIGH byte of a half */
static inline float kfo_f8e5m2_to_f32(uint8_t b) { return kfo_half_to_f32((uint16_t)((uint16_t)b << 8)); }
/* src/g_float.hpp:433-443: float -> half (RNE) then TRUNCATE to the high byte */
static inline uint8_t kfo_f32_to_f8e5m2(float f) { return (uint8_t)(kfo_f32_to_half(f) >> 8); }

/* Portable fp32 exp: Cody-Waite reduction + degree-7 Taylor in Horner/fma form. */
static inline float kfo_expf(float x) {
    if (x > 88.72283f) return INFINITY;
    if (x < -87.33654f) return 0.0f;
    if (x != x) return x;
    float t = x * 1.44269502162933349609375f;         /* log2(e) rounded to f32 */
    float n = (t + 12582912.0f) - 12582912.0f;        /* RNE to integer, |t| < 2^22 */
    float r = fmaf(n, -0.693145751953125f, x);        /* ln2 high part (12 significant bits) */
    r = fmaf(n, -1.428606765330187045e-06f, r);       /* ln2 low part */
    float p = 1.98412701138295233249664306640625e-4f; /* 1/5040 */
    p = fmaf(p, r, 1.38888892251998186111450195312500e-3f); /* 1/720 */
    p = fmaf(p, r, 8.33333376795053482055664062500000e-3f); /* 1/120 */
    p = fmaf(p, r, 4.16666679084300994873046875000000e-2f); /* 1/24 */
    p = fmaf(p, r, 1.66666671633720397949218750000000e-1f); /* 1/6 */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int e = (int)n; /* -126 .. 128 */
    int e1 = e / 2, e2 = e - e1;
    uint32_t u1 = (uint32_t)(e1 + 127) << 23, u2 = (uint32_t)(e2 + 127) << 23;
    float s1, s2;
    memcpy(&s1, &u1, 4);
    memcpy(&s2, &u2, 4);
    return (p * s1) * s2;
}

/* kfo_exp2_parts: 2^y as f * 2^n with n = the integer nearest to y (ties to even) and f = 2^(y - n) in [2^-1/2, 2^1/2] from a fixed degree-7
 * polynomial (Taylor coefficients ln2^k / k!, Horner in fma form; relative error 5e-9).  The decode attention's canonical softmax (kf_oracle.c
 * section 6, mode CANON; koifish_amd/csrc/kf_device.h kf_exp2_parts states the same recipe) keeps the two parts apart: a power of two rescales
 * exactly, so e^(s - m) needs no agreed-upon maximum m between the slices of a sequence. */
static inline void kfo_exp2_parts(float y, float* f, float* n) {
    const float nn = (y + 12582912.0f) - 12582912.0f; /* RNE to integer, |y| < 2^22 (larger |y|: nn = y, r = 0) */
    const float r = y - nn;                            /* exact */
    float p = 1.52527338e-5f;        /* ln2^7 / 5040 */
    p = fmaf(p, r, 1.54035304e-4f);  /* ln2^6 / 720 */
    p = fmaf(p, r, 1.33335581e-3f);  /* ln2^5 / 120 */
    p = fmaf(p, r, 9.61812911e-3f);  /* ln2^4 / 24 */
    p = fmaf(p, r, 5.55041087e-2f);  /* ln2^3 / 6 */
    p = fmaf(p, r, 2.40226507e-1f);  /* ln2^2 / 2 */
    p = fmaf(p, r, 6.93147181e-1f);  /* ln2 */
    p = fmaf(p, r, 1.0f);
    *f = p, *n = nn;
}

/* kfo_logf: the fixed fp32 natural log of the cross-entropy loss (koifish_amd/csrc/kf_device.h kf_logf states the same recipe
 * independently; the reference calls CUDA logf, fused_classifier.cuh:84).  x = m * 2^e, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
 * log m = 2s + 2s*z*P(z), z = s^2.  tests/test_oracle_loss.py pins it to libm logf within 2 ulp. */
static inline float kfo_logf(float x) {
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u;
    memcpy(&u, &x, 4);
    if (u < 0x00800000u) {
        x *= 8388608.0f, e = -23;
        memcpy(&u, &x, 4);
    }
    e += (int)(u >> 23) - 127;
    u = (u & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &u, 4);
    if (m > 1.41421353816986083984375f) m *= 0.5f, e += 1;
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 9.0909093618392944336e-2f;
    p = fmaf(p, z, 1.1111111193895339966e-1f);
    p = fmaf(p, z, 1.4285714924335479736e-1f);
    p = fmaf(p, z, 2.0000000298023223877e-1f);
    p = fmaf(p, z, 3.3333334326744079590e-1f);
    const float s2 = s + s;
    const float fe = (float)e;
    const float lo = fmaf(s2 * z, p, fe * 9.058001351536227e-6f);
    return fmaf(fe, 6.9313812255859375e-1f, s2 + lo);
}

#endif
