// fbsmi_prims.hip -- generic-tier device primitives of libfbsmi and their C ABI (include/fbsmi.h):
// JAX-compatible PRNG draws, canonical-tree cumsum / sum / logsumexp / normalise, searchsorted, the
// unconditional (fbs/samplers/resampling.py) and conditional (fbs/samplers/csmc/resamplings.py)
// resamplers, categorical draw, force_move (fbs/samplers/gibbs.py:171-214), row gather / set,
// ancestor back-trace (fbs/samplers/csmc/csmc.py:262-267).
//
// These serve the closure-driven Python tier (arbitrary user transition / likelihood closures on
// torch tensors); the fused linear-Gaussian sweep lives in fbsmi_lg.hip.  gfx950 only.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fbsmi.h"
#include "fbsmi_device.h"
#include "fbsmi_host.h"

namespace fbsmi {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------
// PRNG kernels
// ------------------------------------------------------------------------------------------
template <int MODE>  // 0 bits, 1 uniform, 2 normal, 3 -log(uniform)
__global__ void k_random(uint32_t k0, uint32_t k1, uint64_t n, void* out) {
    const uint64_t half = (n + 1) >> 1;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < half; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = i + half;
        uint32_t o0, o1;
        threefry2x32(k0, k1, (uint32_t)i, j < n ? (uint32_t)j : 0u, o0, o1);
        if (MODE == 0) {
            ((uint32_t*)out)[i] = o0;
            if (j < n) ((uint32_t*)out)[j] = o1;
        } else if (MODE == 1) {
            ((float*)out)[i] = fbsmi_bits_to_unit(o0);
            if (j < n) ((float*)out)[j] = fbsmi_bits_to_unit(o1);
        } else if (MODE == 2) {
            ((float*)out)[i] = normal_from_bits(o0);
            if (j < n) ((float*)out)[j] = normal_from_bits(o1);
        } else {
            ((float*)out)[i] = -fbsmi_logf(fbsmi_bits_to_unit(o0));
            if (j < n) ((float*)out)[j] = -fbsmi_logf(fbsmi_bits_to_unit(o1));
        }
    }
}

// elements [start, start+count) of the flat draw of n_total elements: what a rank that owns a
// slice of a sharded particle ensemble needs so that its noise equals the unsharded draw
template <int MODE>  // 0 bits, 1 uniform, 2 normal
__global__ void k_random_range(uint32_t k0, uint32_t k1, uint64_t n_total, uint64_t start, uint64_t count, void* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = random_bits_at(k0, k1, n_total, start + i);
        if (MODE == 0) ((uint32_t*)out)[i] = b;
        else if (MODE == 1) ((float*)out)[i] = fbsmi_bits_to_unit(b);
        else ((float*)out)[i] = normal_from_bits(b);
    }
}

__global__ void k_randint(uint32_t k0, uint32_t k1, uint64_t n, int32_t lo, int32_t hi, int32_t* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = randint_at(k0, k1, n, i, lo, hi);
}

__global__ void k_math_map(int op, const float* x, const float* y, int64_t n, float* out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        switch (op) {
            case 0: r = fbsmi_expf(v); break;
            case 1: r = fbsmi_logf(v); break;
            case 2: r = fbsmi_log1pf(v); break;
            case 3: r = fbsmi_erfinvf(v); break;
            case 4: r = fbsmi_sqrtf(v); break;
            case 5: r = v / y[i]; break;
            case 6: r = fbsmi_bits_to_normal(fbsmi_f2u(v)); break;  // the definition
            case 7: r = normal_from_bits(fbsmi_f2u(v)); break;     // the form the kernels use (fbsmi_device.h)
            default: {                                               // x / y the way the log-density kernels divide
                const float b = y[i];
                r = (div_by_in_range(v) && div_by_in_range(b)) ? div_by(v, b, 1.0f / b) : v / b;
            } break;
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------------
// tree kernels
// ------------------------------------------------------------------------------------------
// value of the quantity being reduced / scanned at element e
enum { V_PLAIN = 0, V_EXPSHIFT = 1, V_JPROB = 2, V_FMREST = 3, V_FMALPHA = 4, V_WSQ = 5 };

struct ValArgs {
    const float* x;
    int64_t n;
    // V_EXPSHIFT: exp(x - shift), shift = finite_or_zero(max of partmax)
    // V_JPROB:    (1 - x/wmax)/n with [iref] = jfix (jfix < 0 -> 0)
    // V_FMREST:   force_move rest weights ;  V_FMALPHA: nan->0 (temp*rest/(1-w))
    // V_WSQ:      w * w with w = exp(x - *jfix) (jfix: device scalar holding the logsumexp): the terms of 1 / ESS
    const float* partmax;
    int nbmax;
    int32_t iref;
    const float* jfix;  // device scalar or null
};

template <int VAL>
__device__ __forceinline__ float val_at(const ValArgs& a, int64_t e, float shift_or_wmax) {
    if (e >= a.n) return 0.0f;
    const float v = a.x[e];
    if (VAL == V_PLAIN) return v;
    if (VAL == V_EXPSHIFT) return fbsmi_expf(v - shift_or_wmax);
    if (VAL == V_WSQ) {
        const float w = fbsmi_expf(v - shift_or_wmax);
        return w * w;
    }
    if (VAL == V_JPROB) {
        if (e == a.iref) return a.jfix ? *a.jfix : 0.0f;
        return (1.0f - v / shift_or_wmax) / (float)a.n;
    }
    if (VAL == V_FMREST || VAL == V_FMALPHA) {
        // shift_or_wmax carries w_k
        const float w_k = shift_or_wmax;
        const float temp = 1.0f - w_k;
        float rest;
        if (w_k < 1.0f) rest = (e == a.iref ? 0.0f : v) / temp;
        else rest = (float)(1.0 / (double)a.n);
        if (VAL == V_FMREST) return rest;
        const float al = temp * rest / (1.0f - v);
        return (al != al) ? 0.0f : al;
    }
    return v;
}

template <int VAL>
__device__ __forceinline__ float val_prologue(const ValArgs& a, float* s4) {
    if (VAL == V_EXPSHIFT) return finite_or_zero(top_max(a.partmax, a.nbmax, s4));
    if (VAL == V_JPROB) return top_max(a.partmax, a.nbmax, s4);
    if (VAL == V_FMREST || VAL == V_FMALPHA) return a.x[a.iref];
    if (VAL == V_WSQ) return *a.jfix;
    return 0.0f;
}

template <int ITEMS, int VAL>
__global__ void __launch_bounds__(kBlock) k_part_sum(ValArgs a, float* part) {
    __shared__ float s4[4];
    const float aux = val_prologue<VAL>(a, s4);
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float x[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) x[i] = val_at<VAL>(a, base + i, aux);
    TreePath path;
    const float tot = block_upsweep(chunk_total<ITEMS>(x), path, s4);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

template <int ITEMS>
__global__ void __launch_bounds__(kBlock) k_part_max(const float* x, int64_t n, float* part) {
    __shared__ float s4[4];
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (base + i < n) m = fmaxf(m, x[base + i]);
    m = block_max(m, s4);
    if (threadIdx.x == 0) part[blockIdx.x] = m;
}

// single workgroup: (P, E) of every partial's node and the root (dynamic LDS: nbp floats)
__global__ void __launch_bounds__(kBlock) k_top_scan(const float* part, int nb, float* pref, float* pend,
                                                     float* root) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nbp = next_pow2(nb);
    for (int i = threadIdx.x; i < nbp; i += kBlock) lds[i] = i < nb ? part[i] : 0.0f;
    __syncthreads();
    for (int d = 1; d < nbp; d <<= 1) {
        for (int i = threadIdx.x; i < nbp / (2 * d); i += kBlock) {
            const int r = (i + 1) * 2 * d - 1;
            lds[r] = lds[r - d] + lds[r];
        }
        __syncthreads();
    }
    const float rt = lds[nbp - 1];
    if (threadIdx.x == 0) *root = rt;
    if (pref) {
        for (int b = threadIdx.x; b < nb; b += kBlock) {
            float P = 0.0f, E = rt;
            int pos = 0;
            for (int d = nbp >> 1; d >= 1; d >>= 1) {
                const float t = P + lds[pos + d - 1];
                if (b & d) {
                    P = t;
                    pos += d;
                } else {
                    E = t;
                }
            }
            pref[b] = P;
            pend[b] = E;
        }
    }
}

struct TopRef {
    const float* part;
    const float* pref;  // non-null: read (P, E) / root from memory (written by k_top_scan)
    const float* pend;
    const float* root;
    int nb;
};

__device__ __forceinline__ void top_get(const TopRef& t, int b, float* lds256, float& root, float& P, float& E) {
    if (t.pref) {
        root = *t.root;
        P = t.pref[b];
        E = t.pend[b];
    } else {
        top_tree(t.part, t.nb, b, lds256, root, P, E);
    }
}

template <int ITEMS, int VAL>
__global__ void __launch_bounds__(kBlock) k_scan(ValArgs a, TopRef top, float* out) {
    __shared__ float s4[4];
    __shared__ float s_top[256];
    const float aux = val_prologue<VAL>(a, s4);
    float root, P, E;
    top_get(top, blockIdx.x, s_top, root, P, E);
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float x[ITEMS], c[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) x[i] = val_at<VAL>(a, base + i, aux);
    TreePath path;
    block_upsweep(chunk_total<ITEMS>(x), path, s4);
    block_descend(P, E, path);
    chunk_scan<ITEMS>(x, P, E, c);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (base + i < a.n) out[base + i] = c[i];
}

// 1 workgroup: out[0] = log(root(partsum)) + finite_or_zero(max(partmax))
__global__ void __launch_bounds__(kBlock) k_lse_final(const float* partmax, TopRef top, float* out) {
    __shared__ float s4[4];
    __shared__ float s_top[256];
    const float M = finite_or_zero(top_max(partmax, top.nb, s4));
    float root, P, E;
    top_get(top, 0, s_top, root, P, E);
    if (threadIdx.x == 0) out[0] = fbsmi_logf(root) + M;
}

// 1 workgroup: out[0] = root of partials, optionally transformed: mode 1 -> max(1 - root, 0),
// mode 2 -> clip(root, 0, 1), mode 3 -> 1 / root
__global__ void __launch_bounds__(kBlock) k_root_final(TopRef top, int mode, float* out) {
    __shared__ float s_top[256];
    float root, P, E;
    top_get(top, 0, s_top, root, P, E);
    if (mode == 1) root = fmaxf(1.0f - root, 0.0f);
    if (mode == 2) root = root < 0.0f ? 0.0f : (root > 1.0f ? 1.0f : root);
    if (mode == 3) root = 1.0f / root;
    if (threadIdx.x == 0) out[0] = root;
}

__global__ void k_normalise(const float* lw, int64_t n, const float* lse, int log_space, float* out) {
    const float c = *lse;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = lw[i] - c;
        out[i] = log_space ? v : fbsmi_expf(v);
    }
}

__global__ void k_searchsorted(const float* a, int32_t n, int levels, const float* q, int64_t m, int32_t* out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = searchsorted_left(a, n, levels, q[i]);
}

// ------------------------------------------------------------------------------------------
// resampler finishing kernels
// ------------------------------------------------------------------------------------------
// stratified / systematic (resampling.py:43-51): idx = clip(searchsorted(cdf, (i + u)/n))
// systematic without clip = csmc/resamplings.py:120-125
__global__ void k_strat_search(const float* cdf, int32_t n, int levels, uint32_t k0, uint32_t k1, int systematic,
                               int clip, int32_t* idx) {
    const float u0 = systematic ? uniform_at(k0, k1, 1, 0) : 0.0f;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float u = systematic ? u0 : uniform_at(k0, k1, (uint64_t)n, (uint64_t)i);
        const float q = ((float)i + u) / (float)n;
        int32_t k = searchsorted_left(cdf, n, levels, q);
        if (clip) k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
        idx[i] = k;
    }
}

// multinomial by sorted uniforms (resampling.py:62-68): z = cumsum(-log U) over n+1 draws
__global__ void k_ratio_search(const float* cdf, int32_t n, int levels, const float* z, int32_t* idx) {
    const float zl = z[n];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t k = searchsorted_left(cdf, n, levels, z[i] / zl);
        k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
        idx[i] = k;
    }
}

// jax.random.choice(key, n, (m,), p): out[i] = searchsorted(cdf, cdf[n-1] * (1 - U_i)); then
// optional pin out[j] = i_pin (conditional multinomial, csmc/resamplings.py:34-36)
__global__ void k_choice_search(const float* cdf, int32_t n, int levels, uint32_t k0, uint32_t k1, int64_t m,
                                int32_t pin_j, int32_t pin_i, int32_t* out) {
    const float last = cdf[n - 1];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const float u = uniform_at(k0, k1, (uint64_t)m, (uint64_t)i);
        int32_t k = searchsorted_left(cdf, n, levels, last * (1.0f - u));
        if (i == pin_j) k = pin_i;
        out[i] = k;
    }
}

// killing (resampling.py:92-100 / csmc/resamplings.py:66-86).  keys: (a0,a1) key_1, (b0,b1) key_2,
// (c0,c1) key_3.  conditional: rotate by j - J with J ~ Cat(J_prob) (cdfJ), then idx[j] = i.
__global__ void __launch_bounds__(kBlock) k_killing_finish(const float* w, const float* cdf, const float* cdfJ,
                                                           const float* partmax, int nbmax, int32_t n, int levels,
                                                           uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1,
                                                           uint32_t c0, uint32_t c1, int conditional, int32_t i_ref,
                                                           int32_t j_ref, int32_t* idx) {
    __shared__ float s4[4];
    __shared__ int s_shift;
    const float w_max = top_max(partmax, nbmax, s4);
    if (threadIdx.x == 0) {
        int shift = 0;
        if (conditional) {
            const float u3 = uniform_at(c0, c1, 1, 0);
            const int32_t J = searchsorted_left(cdfJ, n, levels, cdfJ[n - 1] * (1.0f - u3));
            long long s = ((long long)j_ref - J) % n;
            if (s < 0) s += n;
            shift = (int)s;
        }
        s_shift = shift;
    }
    __syncthreads();
    const int shift = s_shift;
    const float last = cdf[n - 1];
    for (int32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
        int32_t src = m - shift;
        if (src < 0) src += n;
        const float ws = w[src];
        const float u1 = uniform_at(a0, a1, (uint64_t)n, (uint64_t)src);
        int32_t a = src;
        if (u1 * w_max >= ws) {
            const float u2 = uniform_at(b0, b1, (uint64_t)n, (uint64_t)src);
            a = searchsorted_left(cdf, n, levels, last * (1.0f - u2));
        }
        if (conditional && m == j_ref) a = i_ref;
        idx[m] = a;
    }
}

// categorical: out[0] = searchsorted(cdf, cdf[n-1] * (1 - U)), U = uniform(key, ())
__global__ void k_categorical_finish(const float* cdf, int32_t n, int levels, uint32_t k0, uint32_t k1,
                                     int32_t* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float u = uniform_at(k0, k1, 1, 0);
        out[0] = searchsorted_left(cdf, n, levels, cdf[n - 1] * (1.0f - u));
    }
}

// force_move tail (gibbs.py:207-212): i ~ Cat(rest) [cdf], u ~ U; accept if u (1 - w_i) < 1 - w_k
__global__ void k_force_move_finish(const float* w, const float* cdf, int32_t n, int levels, int32_t k,
                                    uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, int32_t* out_i) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float u1 = uniform_at(a0, a1, 1, 0);
        const int32_t i = searchsorted_left(cdf, n, levels, cdf[n - 1] * (1.0f - u1));
        const float u = uniform_at(b0, b1, 1, 0);
        const float temp = 1.0f - w[k];
        const bool accept = u * (1.0f - w[i]) < temp;
        out_i[0] = accept ? i : k;
    }
}

// ------------------------------------------------------------------------------------------
// data movement
// ------------------------------------------------------------------------------------------
__global__ void k_gather_rows(const float* src, const int32_t* idx, int64_t n, int64_t d, float* dst) {
    const int64_t total = n * d;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / d, c = e - r * d;
        dst[e] = src[(int64_t)idx[r] * d + c];
    }
}

// rows that are a multiple of 16 bytes: one float4 per lane
__global__ void k_gather_rows4(const float4* src, const int32_t* idx, int64_t n, int64_t d4, float4* dst) {
    const int64_t total = n * d4;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / d4, c = e - r * d4;
        dst[e] = src[(int64_t)idx[r] * d4 + c];
    }
}

__global__ void k_set_row(float* dst, int64_t row, const float* src, int64_t d) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < d; c += (int64_t)gridDim.x * blockDim.x)
        dst[row * d + c] = src[c];
}

// a T-long pointer chase: one lane
__global__ void k_backtrace(const int32_t* As, int32_t T, int32_t n, const int32_t* B_T, int32_t* Bs) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t B = B_T[0];
        Bs[T] = B;
        for (int k = T; k >= 1; --k) {
            B = As[(int64_t)(k - 1) * n + B];
            Bs[k - 1] = B;
        }
    }
}

// two-level logsumexp (include/fbsmi_math.h): per-tile (max, sumexp) ...
template <int ITEMS>
__global__ void __launch_bounds__(kBlock) k_part_lse(const float* x, int64_t n, float* pmax, float* psum) {
    __shared__ float xch[2][4];
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float l[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) l[i] = base + i < n ? x[base + i] : -__builtin_inff();
    float m, sx;
    block_lse_partial<ITEMS>(l, xch[0], xch[1], m, sx);
    if (threadIdx.x == 0) {
        pmax[blockIdx.x] = m;
        psum[blockIdx.x] = sx;
    }
}

// ... combined by one workgroup: out[0] = log(tree-sum_t s_t exp(m_t' - M')) + M'  (dynamic LDS: nbp floats)
__global__ void __launch_bounds__(kBlock) k_lse_top(const float* pmax, const float* psum, int nb, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float s4[4];
    float m = -__builtin_inff();
    for (int i = threadIdx.x; i < nb; i += kBlock) m = fmaxf(m, pmax[i]);
    const float Mp = finite_or_zero(block_max(m, s4));
    const int nbp = next_pow2(nb);
    for (int i = threadIdx.x; i < nbp; i += kBlock)
        lds[i] = i < nb ? psum[i] * fbsmi_expf(finite_or_zero(pmax[i]) - Mp) : 0.0f;
    __syncthreads();
    for (int d = 1; d < nbp; d <<= 1) {
        for (int i = threadIdx.x; i < nbp / (2 * d); i += kBlock) {
            const int r = (i + 1) * 2 * d - 1;
            lds[r] = lds[r - d] + lds[r];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = fbsmi_logf(lds[nbp - 1]) + Mp;
}

// ------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------
static inline int grid_for(int64_t n, int block = 256, int cap = 2048) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

int items_for(int64_t n) { return fbsmi_tile_items(n); }  // the tile rule of include/fbsmi_math.h

Workspace carve(void* ws, int64_t n) {
    Workspace W;
    const int64_t nbmax = (n + 1 + kBlock - 1) / kBlock + 1;
    float* p = (float*)ws;
    W.part0 = p; p += nbmax;
    W.part1 = p; p += nbmax;
    W.pref = p; p += nbmax;
    W.pend = p; p += nbmax;
    W.scal = p; p += 64;
    // keep the big arrays 16-byte aligned
    while (((uintptr_t)p) & 15) ++p;
    const int64_t n4 = ((n + 1 + 3) / 4) * 4;
    W.tmp0 = p; p += n4;
    W.tmp1 = p; p += n4;
    W.tmp2 = p; p += n4;
    return W;
}

#define FBSMI_LAUNCH_CHECK()                                                         \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e_));     \
    } while (0)

#define FBSMI_DISPATCH_ITEMS(items, ...)                     \
    do {                                                     \
        if ((items) == 1) { constexpr int ITEMS = 1; __VA_ARGS__; }        \
        else if ((items) == 4) { constexpr int ITEMS = 4; __VA_ARGS__; }   \
        else { constexpr int ITEMS = 16; __VA_ARGS__; }                    \
    } while (0)

// partial sums of VAL over n elements -> part (nb entries); returns nb
template <int VAL>
static int launch_part_sum(const ValArgs& a, float* part, hipStream_t st) {
    const int items = items_for(a.n);
    const int nb = (int)((a.n + (int64_t)kBlock * items - 1) / ((int64_t)kBlock * items));
    FBSMI_DISPATCH_ITEMS(items, (k_part_sum<ITEMS, VAL><<<nb, kBlock, 0, st>>>(a, part)));
    return nb;
}

static int launch_part_max(const float* x, int64_t n, float* part, hipStream_t st) {
    const int items = items_for(n);
    const int nb = (int)((n + (int64_t)kBlock * items - 1) / ((int64_t)kBlock * items));
    FBSMI_DISPATCH_ITEMS(items, (k_part_max<ITEMS><<<nb, kBlock, 0, st>>>(x, n, part)));
    return nb;
}

// make a TopRef for `part`; runs the separate top-scan kernel when the partials do not fit the
// in-prologue tree
static int make_top(const float* part, int nb, float* pref, float* pend, float* root, hipStream_t st, TopRef* out) {
    out->part = part;
    out->nb = nb;
    out->pref = nullptr;
    out->pend = nullptr;
    out->root = nullptr;
    if (nb > 256) {
        if (nb > kMaxTopLds) return fail(FBSMI_ERR_UNSUPPORTED, "too many elements for the top-level tree");
        k_top_scan<<<1, kBlock, sizeof(float) * next_pow2(nb), st>>>(part, nb, pref, pend, root);
        out->pref = pref;
        out->pend = pend;
        out->root = root;
    }
    return FBSMI_OK;
}

// scan of VAL over n -> out, using W.part0 / W.pref / W.scal[60]
template <int VAL>
static int scan_val(const ValArgs& a, float* out, Workspace& W, hipStream_t st) {
    const int nb = launch_part_sum<VAL>(a, W.part0, st);
    TopRef top;
    int rc = make_top(W.part0, nb, W.pref, W.pend, W.scal + 60, st, &top);
    if (rc) return rc;
    const int items = items_for(a.n);
    FBSMI_DISPATCH_ITEMS(items, (k_scan<ITEMS, VAL><<<nb, kBlock, 0, st>>>(a, top, out)));
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

// root of VAL over n -> out[0] (mode: see k_root_final)
template <int VAL>
static int root_val(const ValArgs& a, int mode, float* out, Workspace& W, hipStream_t st) {
    const int nb = launch_part_sum<VAL>(a, W.part0, st);
    TopRef top;
    int rc = make_top(W.part0, nb, nullptr, nullptr, W.scal + 60, st, &top);
    if (rc) return rc;
    if (top.root) top.pref = top.pend = W.scal + 60;  // any non-null: k_root_final only needs root
    k_root_final<<<1, kBlock, 0, st>>>(top, mode, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

static ValArgs plain(const float* x, int64_t n) {
    ValArgs a{};
    a.x = x;
    a.n = n;
    return a;
}

int logsumexp_impl(const float* x, int64_t n, float* out, Workspace& W, hipStream_t st) {
    const int items = items_for(n);
    const int nb = (int)((n + (int64_t)kBlock * items - 1) / ((int64_t)kBlock * items));
    if (nb > kMaxTopLds) return fail(FBSMI_ERR_UNSUPPORTED, "logsumexp: too many elements for the top-level tree");
    FBSMI_DISPATCH_ITEMS(items, (k_part_lse<ITEMS><<<nb, kBlock, 0, st>>>(x, n, W.part1, W.part0)));
    k_lse_top<<<1, kBlock, sizeof(float) * next_pow2(nb), st>>>(W.part1, W.part0, nb, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

}  // namespace fbsmi

using namespace fbsmi;

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int fbsmi_abi_version(void) { return FBSMI_ABI_VERSION; }

const char* fbsmi_last_error(void) { return g_err.c_str(); }

void fbsmi_key_split(uint32_t k0, uint32_t k1, int num, uint32_t* out_host) {
    for (int r = 0; r < num; ++r) split_at(k0, k1, num, r, out_host[2 * r], out_host[2 * r + 1]);
}

int fbsmi_random_bits(uint32_t k0, uint32_t k1, int64_t n, uint32_t* out, void* stream) {
    if (n < 0 || (n > 0 && !out)) return fail(FBSMI_ERR_ARG, "random_bits: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_random<0><<<grid_for((n + 1) / 2), 256, 0, (hipStream_t)stream>>>(k0, k1, (uint64_t)n, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_uniform(uint32_t k0, uint32_t k1, int64_t n, float* out, void* stream) {
    if (n < 0 || (n > 0 && !out)) return fail(FBSMI_ERR_ARG, "uniform: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_random<1><<<grid_for((n + 1) / 2), 256, 0, (hipStream_t)stream>>>(k0, k1, (uint64_t)n, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_normal(uint32_t k0, uint32_t k1, int64_t n, float* out, void* stream) {
    if (n < 0 || (n > 0 && !out)) return fail(FBSMI_ERR_ARG, "normal: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_random<2><<<grid_for((n + 1) / 2), 256, 0, (hipStream_t)stream>>>(k0, k1, (uint64_t)n, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_random_range(int mode, uint32_t k0, uint32_t k1, int64_t n_total, int64_t start, int64_t count, void* out,
                       void* stream) {
    if (mode < 0 || mode > 2 || n_total < 0 || start < 0 || count < 0 || start + count > n_total || (count > 0 && !out))
        return fail(FBSMI_ERR_ARG, "random_range: bad arguments");
    if (count == 0) return FBSMI_OK;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) k_random_range<0><<<grid_for(count), 256, 0, st>>>(k0, k1, (uint64_t)n_total, (uint64_t)start, (uint64_t)count, out);
    else if (mode == 1) k_random_range<1><<<grid_for(count), 256, 0, st>>>(k0, k1, (uint64_t)n_total, (uint64_t)start, (uint64_t)count, out);
    else k_random_range<2><<<grid_for(count), 256, 0, st>>>(k0, k1, (uint64_t)n_total, (uint64_t)start, (uint64_t)count, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_randint(uint32_t k0, uint32_t k1, int64_t n, int32_t lo, int32_t hi, int32_t* out, void* stream) {
    if (n < 0 || (n > 0 && !out)) return fail(FBSMI_ERR_ARG, "randint: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_randint<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(k0, k1, (uint64_t)n, lo, hi, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_math_map(int op, const float* x, const float* y, int64_t n, float* out, void* stream) {
    if (n < 0 || op < 0 || op > 8 || (n > 0 && (!x || !out)) || ((op == 5 || op == 8) && !y))
        return fail(FBSMI_ERR_ARG, "math_map: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_math_map<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(op, x, y, n, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

size_t fbsmi_workspace_bytes(int64_t n) {
    if (n < 1) n = 1;
    const int64_t nbmax = (n + 1 + kBlock - 1) / kBlock + 1;
    const int64_t n4 = ((n + 1 + 3) / 4) * 4;
    return (size_t)(4 * nbmax + 64 + 4 + 3 * n4) * sizeof(float);
}

#define FBSMI_NEED(cond, msg) \
    if (!(cond)) return fail(FBSMI_ERR_ARG, msg)

int fbsmi_cumsum(const float* x, int64_t n, float* out, void* ws, void* stream) {
    FBSMI_NEED(n >= 0 && (n == 0 || (x && out && ws)), "cumsum: bad arguments");
    if (n == 0) return FBSMI_OK;
    Workspace W = carve(ws, n);
    return scan_val<V_PLAIN>(plain(x, n), out, W, (hipStream_t)stream);
}

int fbsmi_sum(const float* x, int64_t n, float* out, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && x && out && ws, "sum: bad arguments");
    Workspace W = carve(ws, n);
    return root_val<V_PLAIN>(plain(x, n), 0, out, W, (hipStream_t)stream);
}

int fbsmi_logsumexp(const float* x, int64_t n, float* out, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && x && out && ws, "logsumexp: bad arguments");
    Workspace W = carve(ws, n);
    return logsumexp_impl(x, n, out, W, (hipStream_t)stream);
}

int fbsmi_normalise(const float* lw, int64_t n, int log_space, float* out, float* out_lse, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && lw && out && ws, "normalise: bad arguments");
    Workspace W = carve(ws, n);
    float* lse = out_lse ? out_lse : W.scal;
    int rc = logsumexp_impl(lw, n, lse, W, (hipStream_t)stream);
    if (rc) return rc;
    k_normalise<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(lw, n, lse, log_space, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_normalise_ess(const float* lw, int64_t n, int log_space, float* out, float* out_lse, float* out_ess, void* ws,
                        void* stream) {
    FBSMI_NEED(n >= 1 && lw && out && ws, "normalise_ess: bad arguments");
    Workspace W = carve(ws, n);
    float* lse = out_lse ? out_lse : W.scal;
    int rc = logsumexp_impl(lw, n, lse, W, (hipStream_t)stream);
    if (rc) return rc;
    if (out_ess) {   // before `out` is written: out may alias lw
        ValArgs a = plain(lw, n);
        a.jfix = lse;
        rc = root_val<V_WSQ>(a, 3, out_ess, W, (hipStream_t)stream);
        if (rc) return rc;
    }
    k_normalise<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(lw, n, lse, log_space, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_searchsorted(const float* a, int32_t n, const float* q, int64_t m, int32_t* out, void* stream) {
    FBSMI_NEED(n >= 1 && m >= 0 && a && (m == 0 || (q && out)), "searchsorted: bad arguments");
    if (m == 0) return FBSMI_OK;
    k_searchsorted<<<grid_for(m), 256, 0, (hipStream_t)stream>>>(a, n, bisect_levels(n), q, m, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_resample(int kind, const float* w, uint32_t k0, uint32_t k1, int32_t n, int32_t* idx, void* ws,
                   void* stream) {
    FBSMI_NEED(n >= 1 && w && idx && ws && kind >= 0 && kind <= 3, "resample: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Workspace W = carve(ws, n);
    const int levels = bisect_levels(n);
    int rc = scan_val<V_PLAIN>(plain(w, n), W.tmp0, W, st);  // cdf
    if (rc) return rc;
    if (kind == 0 || kind == 1) {
        k_strat_search<<<grid_for(n), 256, 0, st>>>(W.tmp0, n, levels, k0, k1, kind == 1, 1, idx);
    } else if (kind == 2) {
        k_random<3><<<grid_for((n + 2) / 2), 256, 0, st>>>(k0, k1, (uint64_t)n + 1, W.tmp1);
        rc = scan_val<V_PLAIN>(plain(W.tmp1, (int64_t)n + 1), W.tmp1, W, st);
        if (rc) return rc;
        k_ratio_search<<<grid_for(n), 256, 0, st>>>(W.tmp0, n, levels, W.tmp1, idx);
    } else {
        uint32_t ks[6];
        fbsmi_key_split(k0, k1, 3, ks);
        const int nbm = launch_part_max(w, n, W.part1, st);
        k_killing_finish<<<grid_for(n, kBlock), kBlock, 0, st>>>(w, W.tmp0, nullptr, W.part1, nbm, n, levels, ks[0],
                                                                  ks[1], ks[2], ks[3], ks[4], ks[5], 0, 0, 0, idx);
    }
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_cond_resample(int kind, uint32_t k0, uint32_t k1, const float* w, int32_t i, int32_t j, int conditional,
                        int32_t n, int32_t* idx, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && w && idx && ws && kind >= 0 && kind <= 2, "cond_resample: bad arguments");
    if (conditional) FBSMI_NEED(i >= 0 && i < n && j >= 0 && j < n, "cond_resample: i, j out of range");
    hipStream_t st = (hipStream_t)stream;
    Workspace W = carve(ws, n);
    const int levels = bisect_levels(n);
    if (kind == 2) {
        if (conditional) return fail(FBSMI_ERR_UNSUPPORTED, "Not implemented, not used.");
        int rc = scan_val<V_PLAIN>(plain(w, n), W.tmp0, W, st);
        if (rc) return rc;
        k_strat_search<<<grid_for(n), 256, 0, st>>>(W.tmp0, n, levels, k0, k1, 1, 0, idx);
        FBSMI_LAUNCH_CHECK();
        return FBSMI_OK;
    }
    if (kind == 0) {
        int rc = scan_val<V_PLAIN>(plain(w, n), W.tmp0, W, st);
        if (rc) return rc;
        k_choice_search<<<grid_for(n), 256, 0, st>>>(W.tmp0, n, levels, k0, k1, (int64_t)n, conditional ? j : -1, i,
                                                     idx);
        FBSMI_LAUNCH_CHECK();
        return FBSMI_OK;
    }
    // killing
    uint32_t ks[6];
    fbsmi_key_split(k0, k1, 3, ks);
    const int nbm = launch_part_max(w, n, W.part1, st);
    if (conditional) {
        // J_prob total with [i] = 0 -> J_i = max(1 - total, 0) -> scal[0]
        ValArgs a = plain(w, n);
        a.partmax = W.part1;
        a.nbmax = nbm;
        a.iref = i;
        a.jfix = nullptr;
        int rc = root_val<V_JPROB>(a, 1, W.scal, W, st);
        if (rc) return rc;
        a.jfix = W.scal;
        rc = scan_val<V_JPROB>(a, W.tmp1, W, st);  // cdfJ
        if (rc) return rc;
    }
    int rc = scan_val<V_PLAIN>(plain(w, n), W.tmp0, W, st);  // cdf
    if (rc) return rc;
    k_killing_finish<<<grid_for(n, kBlock), kBlock, 0, st>>>(w, W.tmp0, W.tmp1, W.part1, nbm, n, levels, ks[0], ks[1],
                                                              ks[2], ks[3], ks[4], ks[5], conditional ? 1 : 0, i, j,
                                                              idx);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_categorical(uint32_t k0, uint32_t k1, const float* w, int32_t n, int32_t* out, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && w && out && ws, "categorical: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Workspace W = carve(ws, n);
    int rc = scan_val<V_PLAIN>(plain(w, n), W.tmp0, W, st);
    if (rc) return rc;
    k_categorical_finish<<<1, 64, 0, st>>>(W.tmp0, n, bisect_levels(n), k0, k1, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_force_move(uint32_t k0, uint32_t k1, const float* w, int32_t k, int32_t n, int32_t* out_i,
                     float* out_alpha, void* ws, void* stream) {
    FBSMI_NEED(n >= 1 && w && out_i && ws && k >= 0 && k < n, "force_move: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Workspace W = carve(ws, n);
    uint32_t ks[4];
    fbsmi_key_split(k0, k1, 2, ks);
    ValArgs a = plain(w, n);
    a.iref = k;
    int rc = scan_val<V_FMREST>(a, W.tmp0, W, st);
    if (rc) return rc;
    k_force_move_finish<<<1, 64, 0, st>>>(w, W.tmp0, n, bisect_levels(n), k, ks[0], ks[1], ks[2], ks[3], out_i);
    FBSMI_LAUNCH_CHECK();
    if (out_alpha) {
        rc = root_val<V_FMALPHA>(a, 2, out_alpha, W, st);
        if (rc) return rc;
    }
    return FBSMI_OK;
}

int fbsmi_gather_rows(const float* src, const int32_t* idx, int64_t n, int64_t d, float* dst, void* stream) {
    FBSMI_NEED(n >= 0 && d >= 1 && (n == 0 || (src && idx && dst)), "gather_rows: bad arguments");
    if (n == 0) return FBSMI_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
        k_gather_rows4<<<grid_for(n * (d / 4), 256, 8192), 256, 0, st>>>((const float4*)src, idx, n, d / 4, (float4*)dst);
    else
        k_gather_rows<<<grid_for(n * d, 256, 8192), 256, 0, st>>>(src, idx, n, d, dst);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_set_row(float* dst, int64_t row, const float* src, int64_t d, void* stream) {
    FBSMI_NEED(dst && src && d >= 1 && row >= 0, "set_row: bad arguments");
    k_set_row<<<grid_for(d), 256, 0, (hipStream_t)stream>>>(dst, row, src, d);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_backtrace(const int32_t* As, int32_t T, int32_t n, const int32_t* B_T, int32_t* Bs, void* stream) {
    FBSMI_NEED(As && B_T && Bs && T >= 0 && n >= 1, "backtrace: bad arguments");
    k_backtrace<<<1, 64, 0, (hipStream_t)stream>>>(As, T, n, B_T, Bs);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

}  // extern "C"
