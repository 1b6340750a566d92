/* pt_numerics.h — the numeric contract of the pt_* boundary.
 *
 * Every implementation of the boundary in pt_api.h (the HIP engine, the CPU
 * oracle) must draw its random numbers from the counter-based generator
 * defined here, in the dimension order documented in DESIGN.md, and must use
 * the elementary functions defined here wherever the algorithm calls sin /
 * cos / exp / pow.  Together with IEEE-754 +,-,*,/,sqrt evaluated without
 * contraction (-ffp-contract=off on both compilers) this makes every discrete
 * path decision (Russian roulette, reflect-vs-refract, closest-hit ties,
 * aperture rejection) identical on x86 and on gfx950, which is what lets the
 * film be compared at matched seeds (BASELINE.json: L-inf < 1e-4).
 *
 * Why not libm / the device math library: x86 glibc and the ROCm device
 * library round transcendental functions differently in the last bit, and a
 * one-ulp difference in a direction flips a hit/miss decision for about one
 * path in 1e6, i.e. for hundreds of paths per frame.
 *
 * The reference takes these functions from Rust std (f32::sin_cos, exp, powf)
 * through the un-vendored `math` crate (gillett-hernandez/rust_cg_math, no
 * pinned revision; /root/reference/Cargo.toml:50-53) and draws from
 * rand::thread_rng (src/renderer/tiled.rs:344), so neither is reproducible;
 * this header is where this build fixes both.
 *
 * Plain C99 / C++ / HIP.  No state.  All functions are branch-light and use
 * only +,-,*,/ and integer ops, so they compile to the same IEEE operations
 * everywhere.
 */
#ifndef PT_NUMERICS_H
#define PT_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PT_HD static inline
#endif

#ifdef __cplusplus
extern "C++" {
#endif

#define PT_PI 3.14159265358979323846f
#define PT_TAU 6.28318530717958647692f
#define PT_F32_EPSILON 1.1920929e-7f
#define PT_INF (__builtin_inff())

/* ------------------------------------------------------------------ bits */
PT_HD uint32_t pt_f2u(float f) {
    union { float f; uint32_t u; } v; v.f = f; return v.u;
}
PT_HD float pt_u2f(uint32_t u) {
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}
PT_HD uint64_t pt_d2u(double f) {
    union { double f; uint64_t u; } v; v.f = f; return v.u;
}
PT_HD double pt_u2d(uint64_t u) {
    union { double f; uint64_t u; } v; v.u = u; return v.f;
}

PT_HD int pt_isnan(float x) { return x != x; }
PT_HD int pt_isfinite(float x) { return (pt_f2u(x) & 0x7f800000u) != 0x7f800000u; }
PT_HD float pt_abs(float x) { return pt_u2f(pt_f2u(x) & 0x7fffffffu); }
/* Rust f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN. */
PT_HD float pt_signum(float x) {
    if (x != x) return x;
    return pt_u2f(0x3f800000u | (pt_f2u(x) & 0x80000000u));
}
/* Rust f32::max / min: the non-NaN operand if one is NaN. Zero sign: first operand wins on ties. */
PT_HD float pt_max(float a, float b) { return (a >= b || b != b) ? a : b; }
PT_HD float pt_min(float a, float b) { return (a <= b || b != b) ? a : b; }
/* Rust f32::clamp (NaN stays NaN). */
PT_HD float pt_clamp(float x, float lo, float hi) {
    float r = x;
    if (r < lo) r = lo;
    if (r > hi) r = hi;
    return r;
}
/* floor for |x| < 2^31, exact. */
PT_HD float pt_floor(float x) {
    float t = (float)(int32_t)x;
    return (t > x) ? t - 1.0f : t;
}

/* ------------------------------------------------ counter-based generator
 * Philox4x32-10 (Salmon et al., SC'11).  key = (seed_lo, seed_hi), counter =
 * (pixel, sample, dimension block, stream).  One call yields four uniforms. */
typedef struct pt_u32x4 { uint32_t x, y, z, w; } pt_u32x4;

PT_HD pt_u32x4 pt_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                             uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    pt_u32x4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}

/* 24-bit uniform in [0,1), the same mapping rand's Standard f32 uses. */
PT_HD float pt_u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

typedef struct pt_f32x4 { float x, y, z, w; } pt_f32x4;

/* The four uniforms of dimension block `dim` of sample `sample` of pixel `pixel`. */
PT_HD pt_f32x4 pt_draw4(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    pt_u32x4 r = pt_philox4x32(pixel, sample, dim, 0x70617468u /* "path" */,
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    pt_f32x4 o; o.x = pt_u01(r.x); o.y = pt_u01(r.y); o.z = pt_u01(r.z); o.w = pt_u01(r.w);
    return o;
}

/* The same with another Philox tag word: the medium-aware walk's draws (the reference takes them from the thread RNG, utils.rs:773-778,
 * 1036-1041): tag "mdst", block b = the free-flight samples of bounce b for up to four tracked mediums (x, y, z, w); tag "mphs",
 * block b = x, y: the phase-function sample of bounce b. */
#define PT_TAG_MEDIUM_DISTANCE 0x6d647374u
#define PT_TAG_MEDIUM_PHASE 0x6d706873u
PT_HD pt_f32x4 pt_draw4_tagged(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t tag) {
    pt_u32x4 r = pt_philox4x32(pixel, sample, dim, tag, (uint32_t)seed, (uint32_t)(seed >> 32));
    pt_f32x4 o; o.x = pt_u01(r.x); o.y = pt_u01(r.y); o.z = pt_u01(r.z); o.w = pt_u01(r.w);
    return o;
}

/* Dimension-block layout of one camera sample (DESIGN.md "RNG dimensions"):
 *   block 0                : x,y = pixel jitter (tiled.rs:369), z = wavelength (pt.rs:406)
 *   blocks 1..16           : aperture rejection tries, two (x,y),(z,w) per block
 *                            (projective_camera.rs:102-106)
 *   block 32 + b*(1+L)     : bounce b: x,y = BSDF sample (utils.rs:219), z = roulette (utils.rs:319)
 *   block 32 + b*(1+L)+1+l : bounce b, light sample l: x = env/light choice
 *                            (pt.rs:350-353), y,z = light / env sample (pt.rs:365,377)
 * L = light_samples. */
#define PT_DIM_FILM 0u
#define PT_DIM_APERTURE0 1u
#define PT_APERTURE_BLOCKS 16u
#define PT_DIM_BOUNCE0 32u
PT_HD uint32_t pt_dim_bounce(uint32_t bounce, uint32_t light_samples) {
    return PT_DIM_BOUNCE0 + bounce * (1u + light_samples);
}

/* ------------------------------------------------------------ f32 sin/cos
 * Cody-Waite reduction by pi/2 (three constants) and the Cephes sinf/cosf
 * minimax polynomials on [-pi/4, pi/4].  Max error about 1 ulp for |x| < 1e4;
 * arguments here are at most 2*pi. */
PT_HD void pt_sincos(float x, float* s_out, float* c_out) {
    float fk = pt_floor(x * 0.63661977236758134f + 0.5f);
    int32_t k = (int32_t)fk;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    float r2 = r * r;
    float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2
                + 4.166664568298827e-2f) * r2 * r2 - 0.5f * r2 + 1.0f;
    float s, c;
    switch (k & 3) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    *s_out = s; *c_out = c;
}
PT_HD float pt_sin(float x) { float s, c; pt_sincos(x, &s, &c); return s; }
PT_HD float pt_cos(float x) { float s, c; pt_sincos(x, &s, &c); return c; }

/* ---------------------------------------------------------------- f32 exp
 * Cephes expf: n = round(x*log2 e), two-constant reduction, degree-5
 * polynomial, exact scaling by 2^n (two steps when the result is subnormal). */
PT_HD float pt_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return PT_INF;
    if (x < -103.9f) return 0.0f;
    float fn = pt_floor(1.44269504088896341f * x + 0.5f);
    int32_t n = (int32_t)fn;
    x = x - fn * 0.693359375f;
    x = x - fn * -2.12194440e-4f;
    float z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z
              + x + 1.0f;
    if (n > 127) { p = p * 2.0f; n -= 1; }            /* x close to the overflow threshold */
    if (n < -126) { p = p * 5.42101086242752217e-20f; n += 64; } /* 2^-64, subnormal result */
    if (n < -126) return 0.0f;
    return p * pt_u2f((uint32_t)(n + 127) << 23);
}

/* ------------------------------------------------------- f64 exp and log
 * Used where the reference computes in f64 (the CIE colour-matching fit,
 * math::misc::gaussian) and for powf, so that the f32 result of pow is
 * accurate to an ulp even for large exponents.  Error below 4e-16 relative. */
PT_HD double pt_exp64(double x) {
    if (x != x) return x;
    if (x > 709.0) return (double)PT_INF;
    if (x < -745.0) return 0.0;
    double t = x * 1.4426950408889634074 + 0.5;
    int64_t k = (int64_t)t; if ((double)k > t) k -= 1;
    double fk = (double)k;
    double r = x - fk * 6.93147180369123816490e-01;
    r = r - fk * 1.90821492927058770002e-10;
    /* Taylor to r^13, |r| <= 0.347 */
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k < -1022) { p = p * 5.42101086242752217e-20; k += 64; }
    if (k < -1022) return 0.0;
    if (k > 1023) { p = p * 2.0; k -= 1; }
    return p * pt_u2d((uint64_t)(k + 1023) << 52);
}

PT_HD double pt_log64(double x) {
    if (x != x || x < 0.0) return pt_u2d(0x7ff8000000000000ull);
    if (x == 0.0) return -(double)PT_INF;
    uint64_t b = pt_d2u(x);
    int64_t e = (int64_t)((b >> 52) & 0x7ff);
    if (e == 0) { x = x * 18014398509481984.0; b = pt_d2u(x); e = (int64_t)((b >> 52) & 0x7ff) - 54; }
    if (e == 0x7ff) return x;
    e -= 1023;
    double m = pt_u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double fe = (double)e;
    return fe * 6.93147180369123816490e-01 + (2.0 * s * p + fe * 1.90821492927058770002e-10);
}

/* f32::ln and f32::cbrt for the mediums (hg.rs:98, rayleigh.rs:72-75, 104): through the f64 kernels, rounded once. */
PT_HD float pt_ln(float x) { return (float)pt_log64((double)x); }
PT_HD float pt_cbrt(float x) {
    if (x != x || x == 0.0f) return x;
    const double a = x < 0.0f ? -(double)x : (double)x;
    const float r = (float)pt_exp64(pt_log64(a) / 3.0);
    return x < 0.0f ? -r : r;
}

/* x^y for x >= 0 (the only use is |cos|^n, sharp_light.rs:202-204). */
PT_HD float pt_pow(float x, float y) {
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : PT_INF;
    if (x == 1.0f) return 1.0f;
    return (float)pt_exp64((double)y * pt_log64((double)x));
}

/* ------------------------------------------------------ f32 acos / atan2
 * Needed by direction_to_uv (equirect environment lookups).  Cephes
 * atanf-style range reduction and polynomial; acos through atan2. */
PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }

PT_HD float pt_atan_01(float x) {
    /* atan on [0, +inf) */
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else { y = 0.0f; }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
              - 3.33329491539e-1f) * z * x + x);
    return y;
}
PT_HD float pt_atan2(float y, float x) {
    if (x != x || y != y) return x + y;
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = pt_abs(x), ay = pt_abs(y);
    float a;
    if (ax == 0.0f) a = 1.5707963267948966f;
    else a = pt_atan_01(ay / ax);
    if (x < 0.0f) a = PT_PI - a;
    return (y < 0.0f) ? -a : a;
}
PT_HD float pt_acos(float x) {
    if (x != x) return x;
    if (x >= 1.0f) return 0.0f;
    if (x <= -1.0f) return PT_PI;
    float s = pt_sqrt((1.0f - x) * (1.0f + x));
    return pt_atan2(s, x);
}

#ifdef __cplusplus
}
#endif
#endif /* PT_NUMERICS_H */
