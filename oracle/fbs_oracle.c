/*
 * fbs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, plain-C CPU restatement of the zgbkdlm/fbs sampler hot path (reference tree
 * /root/reference; every function cites the reference file:line it follows).  It is the checker
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg run beside the HIP path;
 * nothing in fbs_amd/ may import, link or call it.
 *
 * The reference is pure Python on JAX; its arithmetic (Threefry PRNG, uniform/normal/randint/
 * choice, cumsum, searchsorted, logsumexp) lives in the third-party dependency jax==0.4.26 /
 * jaxlib==0.4.26 (requirements_freeze.txt:43-44), which is absent from /root/reference and not
 * installable here.  This file restates JAX's published algorithms for those primitives.
 *
 * PINNING STATUS
 *   - jax.random.split / random_bits / PRNGKey: pinned bit-for-bit by the reference's only
 *     bit-level fixture, experiments/keys.npy (tests/golden/keys_slice.npy).
 *   - everything floating point (uniform's bit->float map, normal via erf_inv, cumsum order,
 *     choice, logsumexp): PARITY UNPINNED against JAX -- the reference's own tests are
 *     statistical only.  They are pinned here by (a) closed-form statistical known-answer tests
 *     restated from the reference's tests/, (b) scipy/numpy float64 cross-checks.
 *
 * Conventions chosen where XLA leaves the order unspecified (stated once, used everywhere, and
 * matched exactly by the HIP kernels so that GPU <-> oracle comparisons are bit-exact):
 *   - cumsum  = jax.lax.associative_scan order (the CPU lowering of jnp.cumsum): recursive
 *               odd/even pairing; equivalently the Blelloch tree over the index bits.
 *   - sum     = root of the pairwise (index-bit) tree over the input zero-padded to a power of 2.
 *   - exp/log/erf_inv = include/fbsmi_math.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fbsmi_math.h"

#define ORC_API __attribute__((visibility("default")))

/* Particle loops whose iterations are independent may run on several host threads (OpenMP); no
 * reduction is parallelised, so results are bit-identical for any thread count.  This only matters
 * for bench.py's cpu_baseline leg. */
#ifdef _OPENMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(static) if (n >= 4096)")
#else
#define ORC_PARALLEL_FOR
#endif

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* JAX PRNG (jax/_src/prng.py, threefry2x32)                             */
/* ------------------------------------------------------------------------------------------ */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* Threefry-2x32, 20 rounds. */
ORC_API void orc_threefry2x32(const uint32_t key[2], uint32_t c0, uint32_t c1, uint32_t out[2]) {
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int g = 0; g < 5; ++g) {
        const int* rot = R[g & 1];
        for (int r = 0; r < 4; ++r) {
            x0 += x1;
            x1 = rotl32(x1, rot[r]);
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    out[0] = x0;
    out[1] = x1;
}

/* random_bits(key, 32, (n,)), original (non-partitionable) counter layout: counters 0..n-1
 * padded to even, first half on lane 0, second half on lane 1. */
ORC_API void orc_random_bits(const uint32_t key[2], int64_t n, uint32_t* out) {
    const int64_t half = (n + 1) / 2;
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < half; ++i) {
        const int64_t j = i + half;
        uint32_t o[2];
        orc_threefry2x32(key, (uint32_t)i, j < n ? (uint32_t)j : 0u, o);
        out[i] = o[0];
        if (j < n) out[j] = o[1];
    }
}

/* jax.random.split(key, num) -> (num, 2) */
ORC_API void orc_split(const uint32_t key[2], int num, uint32_t* out) {
    orc_random_bits(key, 2 * (int64_t)num, out);
}

/* jax.random.uniform(key, (n,), float32) in [0,1) */
ORC_API void orc_uniform(const uint32_t key[2], int64_t n, float* out) {
    uint32_t* bits = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    orc_random_bits(key, n, bits);
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_bits_to_unit(bits[i]);
    free(bits);
}

/* jax.random.normal(key, (n,), float32) */
ORC_API void orc_normal(const uint32_t key[2], int64_t n, float* out) {
    uint32_t* bits = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    orc_random_bits(key, n, bits);
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_bits_to_normal(bits[i]);
    free(bits);
}

/* jax.random.randint(key, (n,), lo, hi) int32 */
ORC_API void orc_randint(const uint32_t key[2], int64_t n, int32_t lo, int32_t hi, int32_t* out) {
    uint32_t ks[4];
    orc_split(key, 2, ks);
    uint32_t* hb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    uint32_t* lb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    orc_random_bits(ks, n, hb);
    orc_random_bits(ks + 2, n, lb);
    uint32_t span = (uint32_t)(hi - lo);
    if (hi <= lo) span = 1;
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t off = (hb[i] % span) * mult + (lb[i] % span);
        off %= span;
        out[i] = lo + (int32_t)off;
    }
    free(hb);
    free(lb);
}

/* ------------------------------------------------------------------------------------------ */
/* cumsum / sum / searchsorted / logsumexp                                                     */
/* ------------------------------------------------------------------------------------------ */

/* jnp.cumsum on CPU = lax.associative_scan(add): restated literally (recursive odd/even). */
static void assoc_scan(const float* x, int64_t n, float* out) {
    if (n <= 0) return;
    if (n == 1) {
        out[0] = x[0];
        return;
    }
    const int64_t nr = n / 2;
    float* red = (float*)malloc(sizeof(float) * (size_t)nr);
    float* odd = (float*)malloc(sizeof(float) * (size_t)nr);
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < nr; ++i) red[i] = x[2 * i] + x[2 * i + 1];
    assoc_scan(red, nr, odd);
    out[0] = x[0];
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < nr; ++i) out[2 * i + 1] = odd[i];
    const int64_t ne = (n - 1) / 2; /* number of i with 2i + 2 < n */
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < ne; ++i) out[2 * i + 2] = odd[i] + x[2 * i + 2];
    free(red);
    free(odd);
}

ORC_API void orc_cumsum(const float* x, int64_t n, float* out) { assoc_scan(x, n, out); }

/* sum = root of the pairwise tree over x zero-padded to a power of two (x + 0 == x): level by level,
 * node i of the next level = node 2i + node 2i+1 (a missing right node contributes nothing). */
ORC_API float orc_sum(const float* x, int64_t n) {
    if (n <= 0) return 0.0f;
    float* a = (float*)malloc(sizeof(float) * (size_t)n);
    float* b = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(a, x, sizeof(float) * (size_t)n);
    int64_t m = n;
    while (m > 1) {
        const int64_t h = m / 2;
        ORC_PARALLEL_FOR
        for (int64_t i = 0; i < h; ++i) b[i] = a[2 * i] + a[2 * i + 1];
        if (m & 1) b[h] = a[m - 1];
        m = h + (m & 1);
        float* t = a;
        a = b;
        b = t;
    }
    const float r = a[0];
    free(a);
    free(b);
    return r;
}

ORC_API float orc_max(const float* x, int64_t n) {
    float m = x[0];
    for (int64_t i = 1; i < n; ++i) m = x[i] > m ? x[i] : m;
    return m;
}

/* jnp.searchsorted(a, q, side='left', method='scan'): the fixed-length bisection. */
ORC_API int32_t orc_searchsorted(const float* a, int32_t n, float q) {
    int32_t low = 0, high = n;
    int levels = 0;
    while ((1ll << levels) < (long long)n + 1) ++levels;
    for (int l = 0; l < levels; ++l) {
        const int32_t mid = (low + high) / 2;
        const int go_left = q <= a[mid];
        if (go_left) high = mid; else low = mid;
    }
    return high;
}

/* jax.random.choice(key, n, (m,), p=w) with replacement: r = c[-1]*(1-U); searchsorted(c, r). */
ORC_API void orc_choice(const uint32_t key[2], const float* w, int32_t n, int64_t m, int32_t* out) {
    float* c = (float*)malloc(sizeof(float) * (size_t)n);
    float* u = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
    orc_cumsum(w, n, c);
    orc_uniform(key, m, u);
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < m; ++i) out[i] = orc_searchsorted(c, n, c[n - 1] * (1.0f - u[i]));
    free(c);
    free(u);
}

/* jax.scipy.special.logsumexp in the two-level form specified in include/fbsmi_math.h (identical to
 * max / exp / sum / log when the input fits one tile). */
ORC_API float orc_logsumexp(const float* x, int64_t n) {
    const int64_t tile = fbsmi_tile(n);
    const int64_t nt = (n + tile - 1) / tile;
    float* tm = (float*)malloc(sizeof(float) * (size_t)nt);
    float* ts = (float*)malloc(sizeof(float) * (size_t)nt);
    float* e = (float*)malloc(sizeof(float) * (size_t)(tile < n ? tile : n));
    float M = -INFINITY;
    for (int64_t t = 0; t < nt; ++t) {
        const int64_t lo = t * tile, cnt = (lo + tile <= n ? tile : n - lo);
        float m = orc_max(x + lo, cnt);
        M = m > M ? m : M;
        if (!(fabsf(m) <= 3.40282347e+38f)) m = 0.0f;
        for (int64_t i = 0; i < cnt; ++i) e[i] = fbsmi_expf(x[lo + i] - m);
        tm[t] = m;
        ts[t] = orc_sum(e, cnt);
    }
    if (!(fabsf(M) <= 3.40282347e+38f)) M = 0.0f;
    for (int64_t t = 0; t < nt; ++t) ts[t] = ts[t] * fbsmi_expf(tm[t] - M);
    const float S = orc_sum(ts, nt);
    free(tm);
    free(ts);
    free(e);
    return fbsmi_logf(S) + M;
}

/* csmc.normalise (fbs/samplers/csmc/csmc.py:273-292) */
ORC_API void orc_normalise(float* lw, int64_t n, int log_space) {
    const float c = orc_logsumexp(lw, n);
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < n; ++i) {
        lw[i] = lw[i] - c;
        if (!log_space) lw[i] = fbsmi_expf(lw[i]);
    }
}

/* Effective sample size of the unnormalised log-weights lw: 1 / sum_i w_i^2, w_i = exp(lw_i - logsumexp(lw)) in float32,
 * the sum in the canonical pairwise-tree order.  A diagnostic of this build (SURVEY.md 8b `out_ess`): the reference
 * computes no ESS; the definition follows its normalise (csmc.py:289-292) for the weights. */
ORC_API float orc_ess(const float* lw, int64_t n) {
    const float c = orc_logsumexp(lw, n);
    float* q = (float*)malloc(sizeof(float) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const float w = fbsmi_expf(lw[i] - c);
        q[i] = w * w;
    }
    const float s = orc_sum(q, n);
    free(q);
    return 1.0f / s;
}

ORC_API void orc_exp(const float* x, int64_t n, float* out) {
    ORC_PARALLEL_FOR
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_expf(x[i]);
}
ORC_API void orc_log(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_logf(x[i]);
}
ORC_API void orc_erfinv(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_erfinvf(x[i]);
}
ORC_API void orc_log1p(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_log1pf(x[i]);
}
ORC_API void orc_sqrt(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = fbsmi_sqrtf(x[i]);
}
ORC_API void orc_div(const float* x, const float* y, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = x[i] / y[i];
}

/* Exhaustive monotonicity check of fbsmi_expf over every float32 in [lo, hi] (lo <= hi < 0 or
 * 0 <= lo <= hi walked in value order).  Returns the number of violations. */
ORC_API int64_t orc_exp_monotone_violations(float lo, float hi) {
    int64_t bad = 0;
    float x = lo;
    float prev = fbsmi_expf(x);
    while (x < hi) {
        x = nextafterf(x, INFINITY);
        const float y = fbsmi_expf(x);
        if (y < prev) ++bad;
        prev = y;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* Unconditional resamplers (fbs/samplers/resampling.py)                                       */
/* ------------------------------------------------------------------------------------------ */

/* _systematic_or_stratified, resampling.py:43-51 */
static void sys_or_strat(const float* w, const uint32_t key[2], int32_t n, int is_systematic, int32_t* idx) {
    float* c = (float*)malloc(sizeof(float) * (size_t)n);
    float* u = (float*)malloc(sizeof(float) * (size_t)n);
    orc_cumsum(w, n, c);
    if (is_systematic) {
        float u0;
        orc_uniform(key, 1, &u0);
        for (int32_t i = 0; i < n; ++i) u[i] = u0;
    } else {
        orc_uniform(key, n, u);
    }
    for (int32_t i = 0; i < n; ++i) {
        const float q = ((float)i + u[i]) / (float)n;
        int32_t k = orc_searchsorted(c, n, q);
        k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
        idx[i] = k;
    }
    free(c);
    free(u);
}

ORC_API void orc_systematic(const float* w, const uint32_t key[2], int32_t n, int32_t* idx) {
    sys_or_strat(w, key, n, 1, idx); /* resampling.py:54-55 */
}
ORC_API void orc_stratified(const float* w, const uint32_t key[2], int32_t n, int32_t* idx) {
    sys_or_strat(w, key, n, 0, idx); /* resampling.py:58-59 */
}

/* multinomial via sorted uniforms, resampling.py:36-40,62-68 */
ORC_API void orc_multinomial(const float* w, const uint32_t key[2], int32_t n, int32_t* idx) {
    float* us = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* z = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* c = (float*)malloc(sizeof(float) * (size_t)n);
    orc_uniform(key, n + 1, us);
    for (int32_t i = 0; i <= n; ++i) us[i] = -fbsmi_logf(us[i]);
    orc_cumsum(us, n + 1, z);
    orc_cumsum(w, n, c);
    for (int32_t i = 0; i < n; ++i) {
        int32_t k = orc_searchsorted(c, n, z[i] / z[n]);
        k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
        idx[i] = k;
    }
    free(us);
    free(z);
    free(c);
}

/* The shared first half of both killing resamplers (resampling.py:92-100 and
 * csmc/resamplings.py:66-74): key -> (key_1, key_2, key_3); killed_i = U_i*w_max >= w_i;
 * killed slots draw from Cat(w) with key_2. */
static void killing_core(const uint32_t key[2], const float* w, int32_t n, int32_t* idx, uint32_t keys3[6],
                         float* w_max_out) {
    orc_split(key, 3, keys3);
    const float w_max = orc_max(w, n);
    float* u = (float*)malloc(sizeof(float) * (size_t)n);
    int32_t* ch = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    orc_uniform(keys3, n, u);
    orc_choice(keys3 + 2, w, n, n, ch);
    ORC_PARALLEL_FOR
    for (int32_t i = 0; i < n; ++i) {
        const int killed = u[i] * w_max >= w[i];
        idx[i] = killed ? ch[i] : i;
    }
    *w_max_out = w_max;
    free(u);
    free(ch);
}

/* unconditional killing, resampling.py:71-101 */
ORC_API void orc_killing(const float* w, const uint32_t key[2], int32_t n, int32_t* idx) {
    uint32_t k3[6];
    float w_max;
    killing_core(key, w, n, idx, k3, &w_max);
}

/* ------------------------------------------------------------------------------------------ */
/* Conditional resamplers (fbs/samplers/csmc/resamplings.py)                                   */
/* ------------------------------------------------------------------------------------------ */

/* multinomial, csmc/resamplings.py:10-37 */
ORC_API void orc_cond_multinomial(const uint32_t key[2], const float* w, int32_t i, int32_t j, int conditional,
                                  int32_t n, int32_t* idx) {
    orc_choice(key, w, n, n, idx);
    if (conditional) idx[j] = i;
}

/* killing, csmc/resamplings.py:40-88 */
ORC_API void orc_cond_killing(const uint32_t key[2], const float* w, int32_t i, int32_t j, int conditional,
                              int32_t n, int32_t* idx) {
    uint32_t k3[6];
    float w_max;
    killing_core(key, w, n, idx, k3, &w_max);
    if (!conditional) return;
    float* jp = (float*)malloc(sizeof(float) * (size_t)n);
    ORC_PARALLEL_FOR
    for (int32_t m = 0; m < n; ++m) jp[m] = (1.0f - w[m] / w_max) / (float)n;   /* :79 */
    jp[i] = 0.0f;                                                                /* :80 */
    float jpi = 1.0f - orc_sum(jp, n);                                           /* :81 */
    jpi = jpi > 0.0f ? jpi : 0.0f;
    jp[i] = jpi;                                                                 /* :82 */
    int32_t J;
    orc_choice(k3 + 4, jp, n, 1, &J);                                            /* :84 */
    /* jnp.roll(idx, j - J): out[(m + s) mod n] = idx[m]                           :85 */
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(tmp, idx, sizeof(int32_t) * (size_t)n);
    long long s = ((long long)j - J) % n;
    if (s < 0) s += n;
    ORC_PARALLEL_FOR
    for (int32_t m = 0; m < n; ++m) idx[(m + s) % n] = tmp[m];
    idx[j] = i;                                                                  /* :86 */
    free(tmp);
    free(jp);
}

/* systematic, csmc/resamplings.py:91-125 (conditional variant raises NotImplementedError :129;
 * returns -1 for it). No clip in the reference's _standard_systematic. */
ORC_API int orc_cond_systematic(const uint32_t key[2], const float* w, int32_t i, int32_t j, int conditional,
                                int32_t n, int32_t* idx) {
    (void)i;
    (void)j;
    if (conditional) return -1;
    float u0;
    orc_uniform(key, 1, &u0);
    float* c = (float*)malloc(sizeof(float) * (size_t)n);
    orc_cumsum(w, n, c);
    for (int32_t m = 0; m < n; ++m) idx[m] = orc_searchsorted(c, n, ((float)m + u0) / (float)n);
    free(c);
    return 0;
}

/* barker_move, csmc/csmc.py:295-297: one categorical draw. */
ORC_API int32_t orc_categorical(const uint32_t key[2], const float* w, int32_t n) {
    int32_t out;
    orc_choice(key, w, n, 1, &out);
    return out;
}

/* force_move, fbs/samplers/gibbs.py:171-214 */
ORC_API int32_t orc_force_move(const uint32_t key[2], const float* w, int32_t k, int32_t M, float* alpha_out) {
    uint32_t ks[4];
    orc_split(key, 2, ks);
    const float w_k = w[k];
    const float temp = 1.0f - w_k;
    float* rest = (float*)malloc(sizeof(float) * (size_t)M);
    /* threshold = max(1 - exp(-M), 1 - 1e-12) evaluated in float32 (gibbs.py:203) */
    float thr = 1.0f - fbsmi_expf(-(float)M);
    const float thr2 = 1.0f - 1e-12f;
    thr = thr > thr2 ? thr : thr2;
    if (w_k < thr) {
        for (int32_t m = 0; m < M; ++m) rest[m] = (m == k ? 0.0f : w[m]) / temp;
    } else {
        for (int32_t m = 0; m < M; ++m) rest[m] = (float)(1.0 / (double)M);
    }
    int32_t i;
    orc_choice(ks, rest, M, 1, &i);
    float u;
    orc_uniform(ks + 2, 1, &u);
    const int accept = u * (1.0f - w[i]) < temp;
    if (alpha_out) {
        float* a = (float*)malloc(sizeof(float) * (size_t)M);
        for (int32_t m = 0; m < M; ++m) {
            float v = temp * rest[m] / (1.0f - w[m]);
            a[m] = (v != v) ? 0.0f : v; /* nansum */
        }
        float al = orc_sum(a, M);
        al = al < 0.0f ? 0.0f : (al > 1.0f ? 1.0f : al);
        *alpha_out = al;
        free(a);
    }
    free(rest);
    return accept ? i : k;
}

/* ------------------------------------------------------------------------------------------ */
/* Linear-Gaussian model closures (experiments/toy/gp_gibbs.py:73-149, tests/test_gibbs.py:41-92) */
/* reverse drift is affine, f(z, t_k) = G_k z + g_k.                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int32_t du, dv, T;
    float dt;
    const float* G;       /* [T][D][D] row-major; step k uses t_prev = ts[k]          */
    const float* g;       /* [T][D]                                                    */
    const float* sd;      /* [T]  sqrt(dt) * dispersion(T - ts[k])                     */
    const float* lognorm; /* [T]  log(2*pi*sd^2)                                       */
    const float* F;       /* [T]  forward transition ts[k]->ts[k+1]: x' = F x + sqQ xi */
    const float* sqQ;     /* [T]                                                       */
} orc_lg;

/* drift_r = g_r + sum_c G_rc z_c as a c-ordered fma chain started at g_r */
static inline float lg_drift_row(const orc_lg* m, int k, int r, const float* u, const float* v) {
    const int D = m->du + m->dv;
    const float* Gr = m->G + ((size_t)k * D + r) * D;
    float acc = m->g[(size_t)k * D + r];
    for (int c = 0; c < m->du; ++c) acc = fbsmi_fmaf(Gr[c], u[c], acc);
    for (int c = 0; c < m->dv; ++c) acc = fbsmi_fmaf(Gr[m->du + c], v[c], acc);
    return acc;
}

/* jax.scipy.stats.norm.logpdf(x, loc, scale) = (log(2 pi scale^2) + (x-loc)^2/scale^2) / -2 */
static inline float norm_logpdf(float x, float loc, float sd2, float lognorm) {
    const float d = x - loc;
    return (lognorm + (d * d) / sd2) / -2.0f;
}

/* transition_sampler(us_prev, v_prev, t_prev=ts[k], key): gp_gibbs.py:120-122 */
ORC_API void orc_lg_transition_sampler(const orc_lg* m, int k, const float* us_prev, const float* v_prev,
                                       const uint32_t key[2], int32_t n, float* us) {
    const int du = m->du;
    float* xi = (float*)malloc(sizeof(float) * (size_t)n * du);
    orc_normal(key, (int64_t)n * du, xi);
    ORC_PARALLEL_FOR
    for (int32_t p = 0; p < n; ++p) {
        const float* u = us_prev + (size_t)p * du;
        for (int r = 0; r < du; ++r) {
            const float dr = lg_drift_row(m, k, r, u, v_prev);
            us[(size_t)p * du + r] = (u[r] + dr * m->dt) + m->sd[k] * xi[(size_t)p * du + r];
        }
    }
    free(xi);
}

/* likelihood_logpdf(v, us_prev, v_prev, t_prev=ts[k]): gp_gibbs.py:131-135 */
ORC_API void orc_lg_likelihood_logpdf(const orc_lg* m, int k, const float* v, const float* us_prev,
                                      const float* v_prev, int32_t n, float* lw) {
    const int du = m->du, dv = m->dv;
    const float sd2 = m->sd[k] * m->sd[k];
    ORC_PARALLEL_FOR
    for (int32_t p = 0; p < n; ++p) {
        const float* u = us_prev + (size_t)p * du;
        float acc = 0.0f;
        for (int r = 0; r < dv; ++r) {
            const float dr = lg_drift_row(m, k, du + r, u, v_prev);
            const float cond_m = v_prev[r] + dr * m->dt;
            const float lp = norm_logpdf(v[r], cond_m, sd2, m->lognorm[k]);
            acc = r == 0 ? lp : acc + lp;
        }
        lw[p] = acc;
    }
}

/* transition_logpdf(u, us_prev, v_prev, t_prev=ts[k]): gp_gibbs.py:124-129 */
ORC_API void orc_lg_transition_logpdf(const orc_lg* m, int k, const float* u_new, const float* us_prev,
                                      const float* v_prev, int32_t n, float* lw) {
    const int du = m->du;
    const float sd2 = m->sd[k] * m->sd[k];
    for (int32_t p = 0; p < n; ++p) {
        const float* u = us_prev + (size_t)p * du;
        float acc = 0.0f;
        for (int r = 0; r < du; ++r) {
            const float dr = lg_drift_row(m, k, r, u, v_prev);
            const float mean = u[r] + dr * m->dt;
            const float lp = norm_logpdf(u_new[r], mean, sd2, m->lognorm[k]);
            acc = r == 0 ? lp : acc + lp;
        }
        lw[p] = acc;
    }
}

/* simulate_cond_forward(key, x0, ts), keep_path=True: fbs/sdes/linear.py:190-221.
 * x0 has D entries; path is (T+1, D). */
ORC_API void orc_lg_fwd_sampler(const orc_lg* m, const uint32_t key[2], const float* x0, int32_t D, float* path) {
    const int T = m->T;
    float* xi = (float*)malloc(sizeof(float) * (size_t)T * D);
    orc_normal(key, (int64_t)T * D, xi);
    memcpy(path, x0, sizeof(float) * (size_t)D);
    for (int k = 0; k < T; ++k)
        for (int c = 0; c < D; ++c)
            path[(size_t)(k + 1) * D + c] = m->F[k] * path[(size_t)k * D + c] + m->sqQ[k] * xi[(size_t)k * D + c];
    free(xi);
}

/* ------------------------------------------------------------------------------------------ */
/* csmc.forward_pass, fbs/samplers/csmc/csmc.py:80-164, with the LG closures and a selectable   */
/* conditional resampler (0 = killing, 1 = multinomial).                                       */
/* us0 (n,du) is what init_sampler returned (csmc.py:151); lw0 (n) what init_likelihood_logpdf      */
/* returns, or NULL for the explicit_final rule of gibbs.py:136-137 (evaluated after the pin).   */
/* Outputs As (T,n), log_wss (T+1,n), uss (T+1,n,du) may each be NULL (not stored).            */
/* us_last (n,du) and lw_last (n) always receive the final particles / log-weights.           */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_csmc_forward_pass_lg(const orc_lg* m, const uint32_t key[2], const float* us_star,
                                      const int32_t* bs_star, const float* vs, const float* us0,
                                      const float* lw0, int32_t n, int resampler, int32_t* As, float* log_wss,
                                      float* uss, float* us_last, float* lw_last) {
    const int du = m->du, dv = m->dv, T = m->T;
    uint32_t k2[4];
    orc_split(key, 2, k2); /* key_init (unused here: caller drew us0), key_scan  csmc.py:150 */
    float* us_prev = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* us = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* lw = (float*)malloc(sizeof(float) * (size_t)n);
    float* w = (float*)malloc(sizeof(float) * (size_t)n);
    int32_t* A = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(T > 0 ? T : 1));

    memcpy(us, us0, sizeof(float) * (size_t)n * du);
    memcpy(us + (size_t)bs_star[0] * du, us_star, sizeof(float) * (size_t)du); /* csmc.py:152 */
    /* csmc.py:154 init_likelihood_logpdf(vs[0], us0, vs[1]) is evaluated on the PINNED us0.
     * lw0 == NULL selects gibbs.py:136-137: likelihood_logpdf(vs[0], u0s, vs[1], ts[0]). */
    if (lw0) memcpy(lw, lw0, sizeof(float) * (size_t)n);
    else orc_lg_likelihood_logpdf(m, 0, vs, us, vs + dv, n, lw);
    orc_normalise(lw, n, 1);                                                   /* csmc.py:155 */
    if (log_wss) memcpy(log_wss, lw, sizeof(float) * (size_t)n);
    if (uss) memcpy(uss, us, sizeof(float) * (size_t)n * du);

    orc_split(k2 + 2, T, keys);                                                /* csmc.py:157 */
    for (int k = 0; k < T; ++k) {                                              /* scan_body :132-148 */
        uint32_t kk[4];
        orc_split(keys + 2 * k, 2, kk); /* key_resampling, key_transition */
        const float* v = vs + (size_t)(k + 1) * dv;
        const float* v_prev = vs + (size_t)k * dv;
        orc_exp(lw, n, w);
        if (resampler == 0) orc_cond_killing(kk, w, bs_star[k], bs_star[k + 1], 1, n, A);
        else orc_cond_multinomial(kk, w, bs_star[k], bs_star[k + 1], 1, n, A);
        ORC_PARALLEL_FOR
        for (int32_t p = 0; p < n; ++p)                                        /* jnp.take :140 */
            memcpy(us_prev + (size_t)p * du, us + (size_t)A[p] * du, sizeof(float) * (size_t)du);
        orc_lg_transition_sampler(m, k, us_prev, v_prev, kk + 2, n, us);       /* :142 */
        memcpy(us + (size_t)bs_star[k + 1] * du, us_star + (size_t)(k + 1) * du, sizeof(float) * (size_t)du); /* :143 */
        orc_lg_likelihood_logpdf(m, k, v, us_prev, v_prev, n, lw);             /* :145 */
        orc_normalise(lw, n, 1);                                               /* :146 */
        if (As) memcpy(As + (size_t)k * n, A, sizeof(int32_t) * (size_t)n);
        if (log_wss) memcpy(log_wss + (size_t)(k + 1) * n, lw, sizeof(float) * (size_t)n);
        if (uss) memcpy(uss + (size_t)(k + 1) * n * du, us, sizeof(float) * (size_t)n * du);
    }
    memcpy(us_last, us, sizeof(float) * (size_t)n * du);
    memcpy(lw_last, lw, sizeof(float) * (size_t)n);
    free(us_prev);
    free(us);
    free(lw);
    free(w);
    free(A);
    free(keys);
}

/* backward_scanning_pass, csmc.py:230-270: B_T ~ Cat(w_T); B_{k-1} = A_k[B_k]. */
ORC_API void orc_backward_scanning_pass(const uint32_t key[2], const int32_t* As, const float* uss,
                                        const float* log_w_T, int32_t T, int32_t n, int32_t du, float* xs_star,
                                        int32_t* bs) {
    float* w = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(w, log_w_T, sizeof(float) * (size_t)n);
    orc_normalise(w, n, 0);
    int32_t B = orc_categorical(key, w, n);
    bs[T] = B;
    memcpy(xs_star + (size_t)T * du, uss + ((size_t)T * n + B) * du, sizeof(float) * (size_t)du);
    for (int k = T; k >= 1; --k) {
        B = As[(size_t)(k - 1) * n + B];
        bs[k - 1] = B;
        memcpy(xs_star + (size_t)(k - 1) * du, uss + ((size_t)(k - 1) * n + B) * du, sizeof(float) * (size_t)du);
    }
    free(w);
}

/* backward_sampling_pass, csmc.py:167-227, LG transition_logpdf. */
ORC_API void orc_backward_sampling_pass_lg(const orc_lg* m, const uint32_t key[2], const float* vs,
                                           const float* uss, const float* log_wss, int32_t n, float* xs_star,
                                           int32_t* bs) {
    const int du = m->du, dv = m->dv, T = m->T;
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(T + 1));
    float* w = (float*)malloc(sizeof(float) * (size_t)n);
    float* gl = (float*)malloc(sizeof(float) * (size_t)n);
    orc_split(key, T + 1, keys);                                               /* :194 */
    memcpy(w, log_wss + (size_t)T * n, sizeof(float) * (size_t)n);
    orc_normalise(w, n, 0);                                                    /* :200 */
    int32_t B = orc_categorical(keys + 2 * T, w, n);                           /* :201 keys[-1] */
    bs[T] = B;
    float* x_t = xs_star + (size_t)T * du;
    memcpy(x_t, uss + ((size_t)T * n + B) * du, sizeof(float) * (size_t)du);
    /* inps = keys[:-1], uss[-2::-1], log_ws[-2::-1], vs[-2::-1], ts[-2::-1]     :214 */
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s; /* time index of xs_{t}, the "t minus 1" of the body */
        orc_lg_transition_logpdf(m, t, x_t, uss + (size_t)t * n * du, vs + (size_t)t * dv, n, gl); /* :205 */
        const float gmax = orc_max(gl, n);
        for (int32_t p = 0; p < n; ++p) w[p] = (gl[p] - gmax) + log_wss[(size_t)t * n + p];     /* :206-207 */
        orc_normalise(w, n, 0);
        B = orc_categorical(keys + 2 * s, w, n);                                                  /* :209 */
        bs[t] = B;
        x_t = xs_star + (size_t)t * du;
        memcpy(x_t, uss + ((size_t)t * n + B) * du, sizeof(float) * (size_t)du);
    }
    free(keys);
    free(w);
    free(gl);
}

/* ------------------------------------------------------------------------------------------ */
/* gibbs_kernel, fbs/samplers/gibbs.py:68-168, LG closures, marg_y=False.                      */
/* x0 (du), y0 (dv), bs_star (T+1); outputs x0_next (du), us_star_next (T+1,du), bs_next (T+1),*/
/* acc (T+1) bytes.  Optional debug outputs: us_T (n,du), lw_T (n) of the CSMC forward pass.   */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_gibbs_kernel_lg(const orc_lg* m, const uint32_t key[2], const float* x0, const float* y0,
                                 const int32_t* bs_star, int32_t nparticles, int explicit_backward,
                                 int explicit_final, float* x0_next, float* us_star_next, int32_t* bs_next,
                                 uint8_t* acc, float* dbg_us_T, float* dbg_lw_T) {
    const int du = m->du, dv = m->dv, T = m->T, D = du + dv;
    uint32_t k3[6];
    orc_split(key, 3, k3); /* key_fwd, key_csmc, key_bridge  gibbs.py:126 */
    float* xy0 = (float*)malloc(sizeof(float) * (size_t)D);
    float* path = (float*)malloc(sizeof(float) * (size_t)(T + 1) * D);
    float* us = (float*)malloc(sizeof(float) * (size_t)(T + 1) * du);
    float* vs = (float*)malloc(sizeof(float) * (size_t)(T + 1) * dv);
    memcpy(xy0, x0, sizeof(float) * (size_t)du);
    memcpy(xy0 + du, y0, sizeof(float) * (size_t)dv);
    orc_lg_fwd_sampler(m, k3, xy0, D, path);                                  /* :127 */
    for (int k = 0; k <= T; ++k) {                                            /* :128-130 */
        memcpy(us + (size_t)k * du, path + (size_t)(T - k) * D, sizeof(float) * (size_t)du);
        memcpy(vs + (size_t)k * dv, path + (size_t)(T - k) * D + du, sizeof(float) * (size_t)dv);
    }
    const int32_t n = explicit_final ? nparticles + 1 : nparticles; /* csmc.py:151 vs gibbs.py:140-141 */
    float* us0 = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* lw0 = (float*)malloc(sizeof(float) * (size_t)n);
    float* usT = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* lwT = (float*)malloc(sizeof(float) * (size_t)n);

    uint32_t k4[8];
    uint32_t kcsmc2[4];
    const uint32_t* key_csmc_fwd;
    if (explicit_backward) {
        orc_split(k3 + 2, 4, k4);                                              /* :147 */
        key_csmc_fwd = k4;
    } else {
        orc_split(k3 + 2, 2, kcsmc2);                                          /* csmc.py:65 */
        key_csmc_fwd = kcsmc2;
    }
    /* init_sampler / init_likelihood_logpdf, gibbs.py:132-144; key_init = split(key_fwd)[0] */
    if (explicit_final) {
        uint32_t kis[4];
        orc_split(key_csmc_fwd, 2, kis);
        orc_normal(kis, (int64_t)n * du, us0);
        free(lw0);
        lw0 = NULL; /* computed inside forward_pass, after the pin */
    } else {
        for (int32_t p = 0; p < n; ++p) memcpy(us0 + (size_t)p * du, us, sizeof(float) * (size_t)du);
        const float c = (float)(-log((double)nparticles));
        for (int32_t p = 0; p < n; ++p) lw0[p] = c;
    }

    if (explicit_backward) {
        orc_csmc_forward_pass_lg(m, key_csmc_fwd, us, bs_star, vs, us0, lw0, n, 0, NULL, NULL, NULL, usT, lwT); /* :148 */
        float* wT = (float*)malloc(sizeof(float) * (size_t)n);
        orc_exp(lwT, n, wT);
        const int32_t idx = orc_force_move(k4 + 2, wT, bs_star[T], n, NULL);   /* :152 */
        memcpy(xy0, usT + (size_t)idx * du, sizeof(float) * (size_t)du);       /* :154 */
        orc_lg_fwd_sampler(m, k4 + 4, xy0, D, path);                           /* :155 */
        for (int k = 0; k <= T; ++k)
            memcpy(us_star_next + (size_t)k * du, path + (size_t)(T - k) * D, sizeof(float) * (size_t)du);
        orc_randint(k4 + 6, T + 1, 0, nparticles, bs_next);                    /* :156 */
        free(wT);
    } else {
        int32_t* As = (int32_t*)malloc(sizeof(int32_t) * (size_t)T * n);
        float* uss = (float*)malloc(sizeof(float) * (size_t)(T + 1) * n * du);
        orc_csmc_forward_pass_lg(m, key_csmc_fwd, us, bs_star, vs, us0, lw0, n, 0, As, NULL, uss, usT, lwT);
        orc_backward_scanning_pass(kcsmc2 + 2, As, uss, lwT, T, n, du, us_star_next, bs_next); /* :158-166 */
        free(As);
        free(uss);
    }
    memcpy(x0_next, us_star_next + (size_t)T * du, sizeof(float) * (size_t)du); /* :167 */
    for (int k = 0; k <= T; ++k) acc[k] = bs_next[k] != bs_star[k];           /* :168 */
    if (dbg_us_T) memcpy(dbg_us_T, usT, sizeof(float) * (size_t)n * du);
    if (dbg_lw_T) memcpy(dbg_lw_T, lwT, sizeof(float) * (size_t)n);
    free(xy0);
    free(path);
    free(us);
    free(vs);
    free(us0);
    free(lw0);
    free(usT);
    free(lwT);
}

/* Run `nsweeps` Gibbs sweeps with the key chain of tests/test_gibbs.py:115-118
 * (key, subkey = split(key); sweep(subkey)); x0s receives (nsweeps, du). */
ORC_API void orc_gibbs_chain_lg(const orc_lg* m, uint32_t key[2], float* x0, const float* y0, int32_t* bs_star,
                                int32_t nparticles, int explicit_backward, int explicit_final, int32_t nsweeps,
                                float* x0s) {
    const int du = m->du, T = m->T;
    float* us_next = (float*)malloc(sizeof(float) * (size_t)(T + 1) * du);
    int32_t* bs_next = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
    uint8_t* acc = (uint8_t*)malloc((size_t)(T + 1));
    float* x0n = (float*)malloc(sizeof(float) * (size_t)du);
    for (int32_t s = 0; s < nsweeps; ++s) {
        uint32_t kk[4];
        orc_split(key, 2, kk);
        key[0] = kk[0];
        key[1] = kk[1];
        orc_gibbs_kernel_lg(m, kk + 2, x0, y0, bs_star, nparticles, explicit_backward, explicit_final, x0n, us_next,
                            bs_next, acc, NULL, NULL);
        memcpy(x0, x0n, sizeof(float) * (size_t)du);
        memcpy(bs_star, bs_next, sizeof(int32_t) * (size_t)(T + 1));
        if (x0s) memcpy(x0s + (size_t)s * du, x0n, sizeof(float) * (size_t)du);
    }
    free(us_next);
    free(bs_next);
    free(acc);
    free(x0n);
}

/* ------------------------------------------------------------------------------------------ */
/* bootstrap_filter (fbs/samplers/smc.py:9-88) and pmcmc_filter_step (smc.py:115-158), LG       */
/* closures; resampling: 0 stratified, 1 systematic, 2 multinomial, 3 killing.                 */
/* ------------------------------------------------------------------------------------------ */
static void uncond_resample(int which, const float* w, const uint32_t key[2], int32_t n, int32_t* idx) {
    switch (which) {
        case 0: orc_stratified(w, key, n, idx); break;
        case 1: orc_systematic(w, key, n, idx); break;
        case 2: orc_multinomial(w, key, n, idx); break;
        default: orc_killing(w, key, n, idx); break;
    }
}

/* init_samples (n,du) is what init_sampler(key_init, vs[0], n) returned.  filtering (T+1,n,du)
 * may be NULL (return_last=True).  Returns negative log-likelihood estimate (log_nell). */
ORC_API float orc_bootstrap_filter_lg(const orc_lg* m, const uint32_t key[2], const float* vs,
                                      const float* init_samples, int32_t n, int resampling, float* last,
                                      float* filtering) {
    const int du = m->du, dv = m->dv, T = m->T;
    uint32_t k2[4];
    orc_split(key, 2, k2);                                                     /* smc.py:77 */
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(T > 0 ? T : 1));
    orc_split(k2 + 2, T, keys);                                                /* :79 */
    float* us_prev = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* us = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* lw = (float*)malloc(sizeof(float) * (size_t)n);
    int32_t* inds = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(us_prev, init_samples, sizeof(float) * (size_t)n * du);
    if (filtering) memcpy(filtering, init_samples, sizeof(float) * (size_t)n * du);
    float log_nell = 0.0f;
    const float logn = (float)log((double)n);
    for (int k = 0; k < T; ++k) {                                              /* scan_body :58-74 */
        uint32_t kk[4];
        orc_split(keys + 2 * k, 2, kk); /* key_proposal, key_resampling */
        const float* v = vs + (size_t)(k + 1) * dv;
        const float* v_prev = vs + (size_t)k * dv;
        orc_lg_transition_sampler(m, k, us_prev, v_prev, kk, n, us);           /* :63 */
        orc_lg_likelihood_logpdf(m, k, v, us_prev, v_prev, n, lw);             /* :65 */
        const float c = orc_logsumexp(lw, n);                                  /* :66 */
        log_nell = log_nell - (c - logn);                                      /* :67 */
        for (int32_t p = 0; p < n; ++p) lw[p] = fbsmi_expf(lw[p] - c);         /* :68-69 */
        uncond_resample(resampling, lw, kk + 2, n, inds);
        for (int32_t p = 0; p < n; ++p)                                        /* :72 */
            memcpy(us_prev + (size_t)p * du, us + (size_t)inds[p] * du, sizeof(float) * (size_t)du);
        if (filtering) memcpy(filtering + (size_t)(k + 1) * n * du, us_prev, sizeof(float) * (size_t)n * du);
    }
    if (last) memcpy(last, us_prev, sizeof(float) * (size_t)n * du);
    free(keys);
    free(us_prev);
    free(us);
    free(lw);
    free(inds);
    return log_nell;
}

/* pmcmc_filter_step, smc.py:115-158: weight -> resample old -> propagate. Returns log_ell. */
ORC_API float orc_pmcmc_filter_step_lg(const orc_lg* m, const uint32_t key[2], const float* vs, const float* u0s,
                                       int32_t n, int resampling, float* uT) {
    const int du = m->du, dv = m->dv, T = m->T;
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(T > 0 ? T : 1));
    orc_split(key, T, keys);                                                   /* :154 */
    float* us_prev = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* us = (float*)malloc(sizeof(float) * (size_t)n * du);
    float* lw = (float*)malloc(sizeof(float) * (size_t)n);
    int32_t* inds = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(us, u0s, sizeof(float) * (size_t)n * du);
    float log_ell = 0.0f;
    const float logn = (float)log((double)n);
    for (int k = 0; k < T; ++k) {                                              /* scan_body :138-152 */
        uint32_t kk[4];
        orc_split(keys + 2 * k, 2, kk); /* key_proposal, key_resampling */
        const float* v = vs + (size_t)(k + 1) * dv;
        const float* v_prev = vs + (size_t)k * dv;
        orc_lg_likelihood_logpdf(m, k, v, us, v_prev, n, lw);                  /* :144 */
        const float c = orc_logsumexp(lw, n);                                  /* :145 */
        log_ell = (log_ell - logn) + c;                                        /* :146 */
        for (int32_t p = 0; p < n; ++p) lw[p] = fbsmi_expf(lw[p] - c);         /* :147-148 */
        uncond_resample(resampling, lw, kk + 2, n, inds);
        for (int32_t p = 0; p < n; ++p)                                        /* :149 */
            memcpy(us_prev + (size_t)p * du, us + (size_t)inds[p] * du, sizeof(float) * (size_t)du);
        orc_lg_transition_sampler(m, k, us_prev, v_prev, kk, n, us);           /* :150 */
    }
    memcpy(uT, us, sizeof(float) * (size_t)n * du);
    free(keys);
    free(us_prev);
    free(us);
    free(lw);
    free(inds);
    return log_ell;
}

/* bootstrap_backward_smoother, smc.py:91-112 (LG transition_logpdf).  filter_us (T+1,n,du).
 * Reproduces the reference's reuse of the PARENT key for the terminal draw (:108-109). */
ORC_API void orc_backward_smoother_lg(const orc_lg* m, const uint32_t key[2], const float* filter_us,
                                      const float* vs, int32_t n, float* traj) {
    const int du = m->du, dv = m->dv, T = m->T;
    uint32_t k2[4];
    orc_split(key, 2, k2); /* key_last (unused by the reference), key_smoother */
    int32_t iT;
    orc_randint(key, 1, 0, n, &iT); /* choice(key, filter_us[-1], axis=0) without p -> randint(key, (), 0, n) */
    memcpy(traj + (size_t)T * du, filter_us + ((size_t)T * n + iT) * du, sizeof(float) * (size_t)du);
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(T > 0 ? T : 1));
    orc_split(k2 + 2, T, keys);
    float* lw = (float*)malloc(sizeof(float) * (size_t)n);
    const float* u_kp1 = traj + (size_t)T * du;
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s; /* filter_us[-2::-1], vs[-2::-1], ts[-2::-1] */
        orc_lg_transition_logpdf(m, t, u_kp1, filter_us + (size_t)t * n * du, vs + (size_t)t * dv, n, lw); /* :101 */
        const float c = orc_logsumexp(lw, n);                                  /* :103 */
        for (int32_t p = 0; p < n; ++p) lw[p] = fbsmi_expf(lw[p] - c);
        const int32_t i = orc_categorical(keys + 2 * s, lw, n);                /* :104 */
        memcpy(traj + (size_t)t * du, filter_us + ((size_t)t * n + i) * du, sizeof(float) * (size_t)du);
        u_kp1 = traj + (size_t)t * du;
    }
    free(keys);
    free(lw);
}

/* ------------------------------------------------------------------------------------------ */
/* CPU-baseline helper: run `nsweeps` explicit-backward sweeps and return nothing; bench.py    */
/* times the call.                                                                             */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_bench_gibbs_lg(const orc_lg* m, uint32_t seed, const float* x0_in, const float* y0,
                                int32_t nparticles, int32_t nsweeps, float* x0_out) {
    uint32_t key[2] = {0u, seed};
    const int du = m->du, T = m->T;
    float* x0 = (float*)malloc(sizeof(float) * (size_t)du);
    int32_t* bs = (int32_t*)calloc((size_t)(T + 1), sizeof(int32_t));
    memcpy(x0, x0_in, sizeof(float) * (size_t)du);
    orc_gibbs_chain_lg(m, key, x0, y0, bs, nparticles, 1, 0, nsweeps, NULL);
    memcpy(x0_out, x0, sizeof(float) * (size_t)du);
    free(x0);
    free(bs);
}
