/*
 * fbsmi_math.h -- the NUMERIC SPECIFICATION of the few float32 transcendental functions the
 * sampler hot path needs (exp, log, log1p, erf_inv), written only in terms of IEEE-754 basic
 * operations (+ - * / sqrt, round-to-nearest-even, fused multiply-add where written
 * explicitly).  The same text is compiled by hipcc for gfx950 device code and by gcc for the
 * host, so that a device result and a host result are the SAME BITS: that is what lets the
 * ancestor indices of a T-step particle sweep be compared bit-for-bit.
 *
 * Must be compiled with floating-point contraction OFF (-ffp-contract=off): every fused
 * multiply-add is spelled fbsmi_fmaf(); no other a*b+c may be fused.
 *
 * What these functions restate (reference = zgbkdlm/fbs, whose arithmetic lives in JAX/XLA):
 *   fbsmi_erfinvf : XLA's float32 erf_inv expansion (Giles' single-precision polynomial,
 *                   w = -log1p(-x*x), branch w < 5), reached from jax.random.normal, which the
 *                   reference calls at fbs/sdes/linear.py:220, experiments/toy/gp_gibbs.py:122.
 *   fbsmi_expf/logf : exp / log of jax.scipy.special.logsumexp and jnp.exp(log_ws)
 *                   (fbs/samplers/csmc/csmc.py:139,289; fbs/samplers/smc.py:66-69,145-149).
 * XLA's own exp/log polynomials are not reproduced (not inspectable here); results agree with
 * them to a few ulp, which is inside the 1e-5 relative tolerance the path is held to.
 */
#ifndef FBSMI_MATH_H
#define FBSMI_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FBSMI_HD __host__ __device__ __forceinline__
#else
#define FBSMI_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif

#define FBSMI_INF_BITS 0x7f800000u
#define FBSMI_NAN_BITS 0x7fc00000u

FBSMI_HD float fbsmi_u2f(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

FBSMI_HD uint32_t fbsmi_f2u(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}

FBSMI_HD float fbsmi_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

FBSMI_HD float fbsmi_sqrtf(float x) { return __builtin_sqrtf(x); }

/* exp(x).  Evaluated in float64 and rounded ONCE to float32: Cody-Waite reduction
 * x = k ln2 + r, |r| <= ln2/2, degree-11 Taylor polynomial by Horner with fma, exact scaling
 * by 2^k, one conversion.  Results below 2^-126 are flushed to +0 (x < -87.3), so no float32
 * subnormal is ever produced.  The float64 evaluation noise (~1e-16) is far below the spacing
 * of exp over adjacent float32 inputs, so the function is monotone non-decreasing over all of
 * float32 (checked exhaustively by tests/test_math_spec.py): max_i exp(x_i) == exp(max_i x_i)
 * is an identity for it, which the HIP path uses for the killing resampler's w_max. */
FBSMI_HD float fbsmi_expf(float x) {
    /* Straight-line: the special cases are selected at the end (same values as early returns), so that
     * several exps in one basic block can be interleaved by the compiler.  The clamp only keeps the
     * discarded lanes' exponent arithmetic in range. */
    const double xd = __builtin_fmin(__builtin_fmax((double)x, -100.0), 100.0);
    const double k = __builtin_rint(xd * 1.4426950408889634);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, xd);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 2.5052108385441720e-08;                 /* 1/11! */
    p = __builtin_fma(p, r, 2.7557319223985888e-07);   /* 1/10! */
    p = __builtin_fma(p, r, 2.7557319223985893e-06);   /* 1/9!  */
    p = __builtin_fma(p, r, 2.4801587301587302e-05);   /* 1/8!  */
    p = __builtin_fma(p, r, 1.9841269841269841e-04);   /* 1/7!  */
    p = __builtin_fma(p, r, 1.3888888888888889e-03);   /* 1/6!  */
    p = __builtin_fma(p, r, 8.3333333333333332e-03);   /* 1/5!  */
    p = __builtin_fma(p, r, 4.1666666666666664e-02);   /* 1/4!  */
    p = __builtin_fma(p, r, 1.6666666666666666e-01);   /* 1/3!  */
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int64_t ki = (int64_t)k;
    uint64_t sb = (uint64_t)(ki + 1023) << 52;
    double sc;
    __builtin_memcpy(&sc, &sb, 8);
    float y = (float)(p * sc);
    y = x < -87.3f ? 0.0f : y;
    y = x > 88.72283f ? fbsmi_u2f(FBSMI_INF_BITS) : y;
    return x != x ? x + x : y;
}

/* log(x), the classic msun/fdlibm single-precision scheme: x = 2^e m, m in [sqrt(1/2), sqrt(2)),
 * f = m - 1, s = f/(2+f), log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)).  Plain operations in the
 * order written (no fma). */
FBSMI_HD float fbsmi_logf(float x) {
    uint32_t ix = fbsmi_f2u(x);
    int e = 0;
    if (x != x) return x + x;
    if (x < 1.17549435e-38f) {
        if (x == 0.0f) return fbsmi_u2f(0xff800000u);   /* -inf */
        if (x < 0.0f) return fbsmi_u2f(FBSMI_NAN_BITS);
        x *= 33554432.0f;                               /* subnormal: scale by 2^25 */
        e = -25;
        ix = fbsmi_f2u(x);
    }
    if (ix >= FBSMI_INF_BITS) return x;
    ix += 0x3f800000u - 0x3f3504f3u;
    e += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float f = fbsmi_u2f(ix) - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)e;
    return dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
}

/* log(1+y) by Kahan's correction of log(fl(1+y)). */
FBSMI_HD float fbsmi_log1pf(float y) {
    const float u = 1.0f + y;
    if (u == 1.0f) return y;
    return fbsmi_logf(u) * (y / (u - 1.0f));
}

/* erf_inv(x), float32: Giles' polynomial as XLA expands it. */
FBSMI_HD float fbsmi_erfinvf(float x) {
    float w = -fbsmi_log1pf(-x * x);
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = fbsmi_sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    const float ax = x < 0.0f ? -x : x;
    if (ax == 1.0f) return x * 3.40282347e+38f;
    return p * x;
}

/* jax.random.uniform bit->float map: top 23 bits into the mantissa of [1,2), minus 1. */
FBSMI_HD float fbsmi_bits_to_unit(uint32_t bits) {
    return fbsmi_u2f((bits >> 9) | 0x3f800000u) - 1.0f;
}

/* jax.random.normal: u = max(lo, f*(hi-lo)+lo), lo = nextafter(-1,0), hi = 1, (hi-lo) rounds
 * to 2.0f; result sqrt(2)*erf_inv(u). */
FBSMI_HD float fbsmi_bits_to_normal(uint32_t bits) {
    const float lo = -0.99999994f;
    float u = fbsmi_bits_to_unit(bits) * 2.0f + lo;
    u = u < lo ? lo : u;
    return 1.41421354f * fbsmi_erfinvf(u);
}

/* ---- summation / logsumexp layout -----------------------------------------------------------------
 * Sums follow one tree over the element index (see fbs_amd/csrc/fbsmi_device.h).  logsumexp is
 * TWO-LEVEL so that it needs one grid-wide dependency instead of two: the input is cut into tiles of
 * fbsmi_tile(n) consecutive elements; tile t has m_t = max, m_t' = (m_t finite ? m_t : 0) and
 * s_t = tree-sum_i exp(x_i - m_t'); then M = max_t m_t, M' = (M finite ? M : 0),
 *     logsumexp(x) = log( tree-sum_t  s_t * exp(m_t' - M') ) + M'.
 * For n <= fbsmi_tile(n) (one tile) this is exactly jax.scipy.special.logsumexp's
 * max / exp / sum / log; for larger n it differs from it by rounding only. */
FBSMI_HD int fbsmi_tile_items(long long n) { return n <= 131072 ? 1 : (n <= 1048576 ? 4 : 16); }
FBSMI_HD int fbsmi_tile(long long n) { return 256 * fbsmi_tile_items(n); }

#if !defined(__clang__) && defined(__GNUC__)
#pragma GCC pop_options
#endif

#endif /* FBSMI_MATH_H */
