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
    /* quadrant k & 3: (sp, cp), (cp, -sp), (-sp, -cp), (-cp, sp) — as selects (a four-way switch on a per-lane value is four divergent branches on the GPU) */
    const int swap = (k & 1) != 0;
    const float s0 = swap ? cp : sp, c0 = swap ? sp : cp;
    *s_out = (k & 2) ? -s0 : s0;
    *c_out = ((k + 1) & 2) ? -c0 : c0;
}
PT_HD float pt_sin(float x) { float s, c; pt_sincos(x, &s, &c); return s; }
PT_HD float pt_cos(float x) { float s, c; pt_sincos(x, &s, &c); return c; }

/* ---------------------------------------------------------------- f32 exp
 * Cephes expf: n = round(x*log2 e), two-constant reduction, degree-5
 * polynomial, exact scaling by 2^n (two steps when the result is subnormal). */
/* (These routines are written as data flow: the special cases are selected at the end, over a main path that every argument takes — an out-of-range
 * one as 0.  On the GPU an early return or an `if` on a per-lane value is a divergent branch, three scalar instructions and more for every value that
 * leaves it, and the scalar unit is shared by four SIMDs (tools/microbench/salu_issue.hip).  Same results, bit for bit.) */
PT_HD float pt_exp(float x0) {
    const int is_nan = x0 != x0, over = x0 > 88.72283905206835f, under = x0 < -103.9f;
    float x = (is_nan | over | under) ? 0.0f : x0;
    float fn = pt_floor(1.44269504088896341f * x + 0.5f);
    int32_t n = (int32_t)fn;
    x = x - fn * 0.693359375f;
    x = x - fn * -2.12194440e-4f;
    float z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z
              + x + 1.0f;
    const int hi = n > 127;                             /* x close to the overflow threshold */
    p = hi ? p * 2.0f : p; n = hi ? n - 1 : n;
    const int sub = n < -126;                           /* 2^-64, subnormal result */
    p = sub ? p * 5.42101086242752217e-20f : p; n = sub ? n + 64 : n;
    const int gone = n < -126;
    float r = p * pt_u2f((uint32_t)((gone ? 0 : n) + 127) << 23);
    r = (gone | under) ? 0.0f : r;
    r = over ? PT_INF : r;
    return is_nan ? x0 : r;
}

/* ------------------------------------------------------- f64 exp and log
 * Used where the reference computes in f64 (the CIE colour-matching fit,
 * math::misc::gaussian) and for powf, so that the f32 result of pow is
 * accurate to an ulp even for large exponents.  Error below 4e-16 relative. */
PT_HD double pt_exp64(double x0) {
    const int is_nan = x0 != x0, over = x0 > 709.0, under = x0 < -745.0;
    const double x = (is_nan | over | under) ? 0.0 : x0;
    double t = x * 1.4426950408889634074 + 0.5;
    int32_t k = (int32_t)t; if ((double)k > t) k -= 1;   /* floor; |t| < 1100 */
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
    const int sub = k < -1022;
    p = sub ? p * 5.42101086242752217e-20 : p; k = sub ? k + 64 : k;
    const int gone = k < -1022;
    const int hi = k > 1023;
    p = hi ? p * 2.0 : p; k = hi ? k - 1 : k;
    double v = p * pt_u2d((uint64_t)(uint32_t)((gone ? 0 : k) + 1023) << 52);
    v = (gone | under) ? 0.0 : v;
    v = over ? (double)PT_INF : v;
    return is_nan ? x0 : v;
}

PT_HD double pt_log64(double x0) {
    const int bad = (x0 != x0) | (x0 < 0.0), zero = x0 == 0.0;
    double x = x0;
    uint64_t b = pt_d2u(x);
    int32_t e = (int32_t)((b >> 52) & 0x7ff);
    const int tiny = e == 0;                      /* subnormal (or zero: selected away below): scaled by 2^54 */
    x = tiny ? x * 18014398509481984.0 : x; b = pt_d2u(x); e = tiny ? (int32_t)((b >> 52) & 0x7ff) - 54 : e;
    const int unbounded = e == 0x7ff;             /* +inf (a NaN is `bad`): returns itself */
    e -= 1023;
    double m = pt_u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    const int upper = m > 1.41421356237309504880;
    m = upper ? m * 0.5 : m; e = upper ? e + 1 : e;
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
    double r = fe * 6.93147180369123816490e-01 + (2.0 * s * p + fe * 1.90821492927058770002e-10);
    r = unbounded ? x0 : r;
    r = zero ? -(double)PT_INF : r;
    return bad ? pt_u2d(0x7ff8000000000000ull) : r;
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
    const int is_nan = (x != x) | (y != y);
    float r = (float)pt_exp64((double)y * pt_log64((double)x));
    r = (x == 1.0f) ? 1.0f : r;
    r = (x == 0.0f) ? ((y > 0.0f) ? 0.0f : PT_INF) : r;
    r = (y == 0.0f) ? 1.0f : r;
    return is_nan ? x + y : r;
}

/* ------------------------------------------------------ f32 acos / atan2
 * Needed by direction_to_uv (equirect environment lookups).  Cephes
 * atanf-style range reduction and polynomial; acos through atan2. */
PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }

PT_HD float pt_atan_01(float x) {
    /* atan on [0, +inf): the three ranges as one division, -1 / x, (x - 1) / (x + 1), x / 1 (the last one exact: x itself) */
    const int far = x > 2.414213562373095f, mid = x > 0.4142135623730950f;
    float y = far ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
    const float num = far ? -1.0f : (mid ? x - 1.0f : x), den = far ? x : (mid ? x + 1.0f : 1.0f);
    x = num / den;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
              - 3.33329491539e-1f) * z * x + x);
    return y;
}
PT_HD float pt_atan2(float y, float x) {
    const int is_nan = (x != x) | (y != y), origin = (x == 0.0f) & (y == 0.0f);
    float ax = pt_abs(x), ay = pt_abs(y);
    /* ax == 0 (and ay != 0): ay / 0 = inf, whose arctangent comes out of pt_atan_01 as pi/2 + (-0) = pi/2, the constant the branch returned */
    float a = pt_atan_01(ay / ax);
    a = (x < 0.0f) ? PT_PI - a : a;
    a = (y < 0.0f) ? -a : a;
    a = origin ? 0.0f : a;
    return is_nan ? x + y : a;
}
PT_HD float pt_acos(float x) {
    const int inside = (x < 1.0f) & (x > -1.0f);   /* (a NaN is not: it returns itself below) */
    const float xs = inside ? x : 0.0f;
    float s = pt_sqrt((1.0f - xs) * (1.0f + xs));
    float r = pt_atan2(s, xs);
    r = (x >= 1.0f) ? 0.0f : r;
    r = (x <= -1.0f) ? PT_PI : r;
    return (x != x) ? x : r;
}

#ifdef __cplusplus
}
#endif
#endif /* PT_NUMERICS_H */
