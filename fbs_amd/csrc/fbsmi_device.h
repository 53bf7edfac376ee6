// fbsmi_device.h -- device-side building blocks shared by every kernel of libfbsmi (gfx950).
//
//  * Threefry-2x32 and the JAX counter layout (random_bits / uniform / normal at a flat index),
//    restating jax/_src/prng.py as the reference reaches it through jax.random.* (call sites:
//    SURVEY.md Appendix A).
//  * The canonical summation tree.  jnp.cumsum on CPU lowers to lax.associative_scan, whose
//    result for element k is the left fold, from the largest block down, of the pairwise-tree
//    sums of the aligned power-of-two blocks that decompose [0, k].  That is exactly a Blelloch
//    up-sweep / down-sweep over the bits of the element index, so the hierarchy
//        items-in-thread (2^a) -> 64 lanes -> 4 waves -> workgroups (top tree)
//    reproduces it bit for bit whatever ITEMS / grid is used (two-value descent, see below).  Sums
//    (logsumexp, J_prob total) are the root of the same tree over the zero-padded input.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fbsmi_math.h"

namespace fbsmi {

constexpr int kBlock = 256;  // threads per workgroup in every tree kernel: 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxTopLds = 16384;  // most workgroup partials one top-level tree may hold in LDS

// ------------------------------------------------------------------------------------------
// PRNG
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                      uint32_t& o0, uint32_t& o1) {
    const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
    uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define FBSMI_TF_ROUND(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
    FBSMI_TF_ROUND(13) FBSMI_TF_ROUND(15) FBSMI_TF_ROUND(26) FBSMI_TF_ROUND(6)
    x0 += k1; x1 += k2 + 1u;
    FBSMI_TF_ROUND(17) FBSMI_TF_ROUND(29) FBSMI_TF_ROUND(16) FBSMI_TF_ROUND(24)
    x0 += k2; x1 += k0 + 2u;
    FBSMI_TF_ROUND(13) FBSMI_TF_ROUND(15) FBSMI_TF_ROUND(26) FBSMI_TF_ROUND(6)
    x0 += k0; x1 += k1 + 3u;
    FBSMI_TF_ROUND(17) FBSMI_TF_ROUND(29) FBSMI_TF_ROUND(16) FBSMI_TF_ROUND(24)
    x0 += k1; x1 += k2 + 4u;
    FBSMI_TF_ROUND(13) FBSMI_TF_ROUND(15) FBSMI_TF_ROUND(26) FBSMI_TF_ROUND(6)
    x0 += k2; x1 += k0 + 5u;
#undef FBSMI_TF_ROUND
    o0 = x0;
    o1 = x1;
}

// element i of jax's random_bits(key, 32, (n,)): counters 0..n-1 padded to even, first half on
// lane 0 of the block cipher, second half on lane 1.
__host__ __device__ __forceinline__ uint32_t random_bits_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i) {
    // one block-cipher call whichever half i is in (no divergent double execution when a wave straddles n/2)
    const uint64_t half = (n + 1) >> 1;
    const bool first = i < half;
    const uint64_t a = first ? i : i - half, b = a + half;
    uint32_t o0, o1;
    threefry2x32(k0, k1, (uint32_t)a, b < n ? (uint32_t)b : 0u, o0, o1);
    return first ? o0 : o1;
}

// elements i (< n/2) and i + n/2 of the same draw, n even: they are the two output words of ONE block-cipher call
__host__ __device__ __forceinline__ void random_bits_pair(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i, uint32_t& lo,
                                                          uint32_t& hi) {
    threefry2x32(k0, k1, (uint32_t)i, (uint32_t)(i + (n >> 1)), lo, hi);
}

__host__ __device__ __forceinline__ float uniform_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i) {
    return fbsmi_bits_to_unit(random_bits_at(k0, k1, n, i));
}

// ------------------------------------------------------------------------------------------
// jax.random.normal from one random word, device form.  fbsmi_bits_to_normal (include/fbsmi_math.h)
// is the definition; only the top 23 bits of the word enter it, so it has 2^23 distinct arguments,
// all of which keep log's argument 1 - u^2 in [2^-23, 1] (a positive normal float) and |u| < 1.
// normal_from_bits() is the same sequence of float32 operations with what cannot happen on that
// domain taken out (the NaN / zero / negative / subnormal / infinity cases of log, the |x| == 1 case
// of erf_inv) and the two quotients formed by v_rcp_f32 plus fused corrections instead of the
// range-proof v_div_scale / v_div_fmas / v_div_fixup sequence.  It returns the SAME BITS as the
// definition for every one of the 2^23 arguments: tests/test_gpu_primitives.py checks all of them on
// the device against the definition evaluated on the device and on the host.  ~75 vector instructions
// against ~150, and no scalar branching: the kernels that draw noise are bound by instruction issue.
// ------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// a / b for b of moderate exponent: reciprocal estimate, one Newton step, quotient, two residual
// corrections (the last one produces the correctly rounded quotient whenever no scaling is needed)
__device__ __forceinline__ float div_lean(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float rem = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(rem, r, q);
}

// a / b for many a and one b, rb = 1.0f / b correctly rounded (Markstein): two rounds of residual correction give
// the correctly rounded quotient as long as nothing underflows or overflows on the way -- div_by_in_range(a) for
// moderate b (the caller's b is a variance, sd^2 with sd = sqrt(dt) * dispersion); 5 instructions against the 11 of
// the range-proof sequence, one of them v_rcp_f32.  tests/test_gpu_primitives.py compares it with `/` on 2^26 pairs.
__device__ __forceinline__ float div_by(float a, float b, float rb) {
    float q = a * rb;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, rb, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, rb, q);
}
// |a| in [2^-60, 2^60]
__device__ __forceinline__ bool div_by_in_range(float a) {
    return ((fbsmi_f2u(a) & 0x7fffffffu) - 0x21800000u) <= (0x5d800000u - 0x21800000u);
}

// the same with ONE residual correction: enough for the two quotients of normal_from_bits on all of its 2^23 arguments
// (the exhaustive test is the proof; not a general-purpose division)
__device__ __forceinline__ float div_lean1(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float q = a * r;
    const float rem = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(rem, r, q);
}

__device__ __forceinline__ float normal_from_bits(uint32_t bits) {
    const float lo = -0.99999994f;
    float x = fbsmi_bits_to_unit(bits) * 2.0f + lo;            // fbsmi_bits_to_normal
    x = x < lo ? lo : x;
    const float y = -x * x;                                      // fbsmi_erfinvf: w = -log1p(-x*x)
    const float u = 1.0f + y;                                    // fbsmi_log1pf
    uint32_t ix = fbsmi_f2u(u) + (0x3f800000u - 0x3f3504f3u);   // fbsmi_logf on a positive normal float
    const int e = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float f = fbsmi_u2f(ix) - 1.0f;
    const float s = div_lean1(f, 2.0f + f);
    const float z = s * s;
    const float w4 = z * z;
    const float t1 = w4 * (0.40000972152f + w4 * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w4 * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)e;
    const float lg = dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
    const float l1p = u == 1.0f ? y : lg * div_lean1(y, u - 1.0f);
    float w = -l1p;
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
    return 1.41421354f * (p * x);
}
#else  // host pass of hipcc: never executed, kept so that device functions parse
__host__ __device__ __forceinline__ float normal_from_bits(uint32_t bits) { return fbsmi_bits_to_normal(bits); }
__host__ __device__ __forceinline__ float div_by(float a, float b, float) { return a / b; }
__host__ __device__ __forceinline__ bool div_by_in_range(float) { return false; }
#endif

__host__ __device__ __forceinline__ float normal_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i) {
    return normal_from_bits(random_bits_at(k0, k1, n, i));
}

// jax.random.split(key, num)[r] -> (out0, out1)
__host__ __device__ __forceinline__ void split_at(uint32_t k0, uint32_t k1, int num, int r, uint32_t& a, uint32_t& b) {
    a = random_bits_at(k0, k1, 2ull * num, 2ull * r);
    b = random_bits_at(k0, k1, 2ull * num, 2ull * r + 1);
}

// jax.random.randint(key, (n,), lo, hi)[i], int32
__host__ __device__ __forceinline__ int32_t randint_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i, int32_t lo,
                                                       int32_t hi) {
    uint32_t a0, a1, b0, b1;
    split_at(k0, k1, 2, 0, a0, a1);
    split_at(k0, k1, 2, 1, b0, b1);
    const uint32_t hb = random_bits_at(a0, a1, n, i);
    const uint32_t lb = random_bits_at(b0, b1, n, i);
    uint32_t span = (uint32_t)(hi - lo);
    if (hi <= lo) span = 1;
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    uint32_t off = (hb % span) * mult + (lb % span);
    off %= span;
    return lo + (int32_t)off;
}

// ------------------------------------------------------------------------------------------
// searchsorted: jnp.searchsorted(a, q, side='left', method='scan') -- the fixed-length bisection,
// reproduced step for step because a tree-summed float32 CDF need not be monotone to the ulp.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int bisect_levels(int n) {
    int l = 0;
    while ((1ll << l) < (long long)n + 1) ++l;
    return l;
}

__device__ __forceinline__ int searchsorted_left(const float* __restrict__ a, int n, int levels, float q) {
    int low = 0, high = n;
    for (int l = 0; l < levels; ++l) {
        const int mid = (low + high) >> 1;
        const bool go_left = q <= a[mid];
        high = go_left ? mid : high;
        low = go_left ? low : mid;
    }
    return high;
}

// ------------------------------------------------------------------------------------------
// cross-lane partners for butterfly trees.  At level m every lane of an aligned 2^m-lane block
// holds the same value, so "the value held by the sibling block" can be fetched from ANY lane of
// that block: DPP quad permutes / row mirrors and the gfx950 row swaps (v_permlane16_swap,
// v_permlane32_swap) do it in a few cycles, where __shfl_xor costs an LDS-crossbar round trip.
// All 64 lanes must be active.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int M>
__device__ __forceinline__ float tree_partner(float v) {
    if constexpr (M == 1) return dpp_mov<0xB1>(v);         // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return dpp_mov<0x141>(v);   // row_half_mirror: lane i <-> 7-i
    else if constexpr (M == 8) return dpp_mov<0x140>(v);   // row_mirror: lane i <-> 15-i
    else if constexpr (M == 16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return __builtin_bit_cast(float, ((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
    } else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __builtin_bit_cast(float, ((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
    }
}

// max over the 64 lanes (every lane receives it)
__device__ __forceinline__ float wave_max(float m) {
    m = fmaxf(m, tree_partner<1>(m));
    m = fmaxf(m, tree_partner<2>(m));
    m = fmaxf(m, tree_partner<4>(m));
    m = fmaxf(m, tree_partner<8>(m));
    m = fmaxf(m, tree_partner<16>(m));
    m = fmaxf(m, tree_partner<32>(m));
    return m;
}

// ------------------------------------------------------------------------------------------
// canonical tree inside one workgroup of kBlock threads
//
// Inclusive scan in lax.associative_scan order = a two-value descent of the index-bit tree.
// Every node (aligned block) carries P = the canonical prefix of everything in front of it and
// E = the canonical inclusive prefix at its last element.  For a node with children L, R:
//        t = P(node) + sum(L);   P(L) = P(node), E(L) = t;   P(R) = t, E(R) = E(node).
// The scan value of element k is E(leaf k); E(root) is the tree sum.  (An "empty" P is 0.0f:
// 0 + x is exact.)
// ------------------------------------------------------------------------------------------
struct TreePath {
    float ls[8];   // sibling sums: [0..5] lane levels (partner's block), [6] other wave of the pair, [7] waves 0+1
    float own[7];  // own block sums before combining: [0..5] lane levels, [6] own wave total
    float hi;      // waves 2+3
};

#define FBSMI_TREE_LEVEL(m, M)                          \
    {                                                   \
        const float o_ = tree_partner<M>(s);            \
        path.ls[m] = o_;                                \
        path.own[m] = s;                                \
        s = ((lane >> m) & 1) ? o_ + s : s + o_;        \
    }

// lane levels of the up-sweep: returns the wave total (every lane)
__device__ __forceinline__ float wave_upsweep(float s, TreePath& path) {
    const int lane = threadIdx.x & 63;
    FBSMI_TREE_LEVEL(0, 1)
    FBSMI_TREE_LEVEL(1, 2)
    FBSMI_TREE_LEVEL(2, 4)
    FBSMI_TREE_LEVEL(3, 8)
    FBSMI_TREE_LEVEL(4, 16)
    FBSMI_TREE_LEVEL(5, 32)
    return s;
}

// wave levels, from the four wave totals
__device__ __forceinline__ float waves_combine(float w0, float w1, float w2, float w3, float mine, TreePath& path) {
    const int wave = threadIdx.x >> 6;
    const float s01 = w0 + w1, s23 = w2 + w3;
    path.own[6] = mine;
    path.ls[6] = (wave & 2) ? ((wave & 1) ? w2 : w3) : ((wave & 1) ? w0 : w1);
    path.ls[7] = s01;
    path.hi = s23;
    return s01 + s23;
}

// Up-sweep. `s` is the tree sum of this thread's own chunk.  Returns the tile total to every
// thread.  `lds4` is 4 floats of LDS scratch (reusable after the call returns).
__device__ __forceinline__ float block_upsweep(float s, TreePath& path, float* lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s = wave_upsweep(s, path);
    __syncthreads();  // protect lds4 against a previous use
    if (lane == 0) lds4[wave] = s;
    __syncthreads();
    return waves_combine(lds4[0], lds4[1], lds4[2], lds4[3], s, path);
}

// K independent up-sweeps sharing ONE LDS exchange (one barrier pair).  `lds` holds 4*K floats
// that no thread reads or writes between the previous barrier and this call.
template <int K>
__device__ __forceinline__ void block_upsweep_n(float (&s)[K], TreePath (&path)[K], float* lds, float (&tot)[K]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = wave_upsweep(s[k], path[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[4 * k + wave] = s[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k)
        tot[k] = waves_combine(lds[4 * k], lds[4 * k + 1], lds[4 * k + 2], lds[4 * k + 3], s[k], path[k]);
}

// Descent from the tile node (P, E) to this thread's chunk node.
__device__ __forceinline__ void block_descend(float& P, float& E, const TreePath& path) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float t = P + path.ls[7];  // left child of the tile = waves 0,1
    if (wave & 2) P = t; else E = t;
    t = P + ((wave & 1) ? path.ls[6] : path.own[6]);  // left child of the wave pair = its even wave
    if (wave & 1) P = t; else E = t;
#pragma unroll
    for (int m = 5; m >= 0; --m) {
        const bool right = (lane >> m) & 1;
        t = P + (right ? path.ls[m] : path.own[m]);
        if (right) P = t; else E = t;
    }
}

// In-thread tree over ITEMS (power of two) consecutive values.
template <int ITEMS>
__device__ __forceinline__ float chunk_total(const float (&x)[ITEMS]) {
    float t[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) t[i] = x[i];
#pragma unroll
    for (int d = 1; d < ITEMS; d <<= 1)
#pragma unroll
        for (int i = 0; i < ITEMS; i += 2 * d) t[i] = t[i] + t[i + d];
    return t[0];
}

template <int ITEMS>
struct ILog2 {
    static constexpr int value = 1 + ILog2<ITEMS / 2>::value;
};
template <>
struct ILog2<1> {
    static constexpr int value = 0;
};

// Given the chunk node's (P, E): the scan values c[i] of its ITEMS leaves.
template <int ITEMS>
__device__ __forceinline__ void chunk_scan(const float (&x)[ITEMS], float P, float E, float (&c)[ITEMS]) {
    constexpr int LG = ILog2<ITEMS>::value;
    float pre[ITEMS], fin[ITEMS];  // per node, stored at the node's first leaf
    pre[0] = P;
    fin[0] = E;
    if (ITEMS > 1) {
        float t[ITEMS];
        float lvl[LG > 0 ? LG : 1][ITEMS];  // lvl[k][i] = sum of the aligned block of 2^k leaves starting at i
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) t[i] = x[i];
#pragma unroll
        for (int k = 0; k < LG; ++k) {
            const int d = 1 << k;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) lvl[k][i] = t[i];
#pragma unroll
            for (int i = 0; i < ITEMS; i += 2 * d) t[i] = t[i] + t[i + d];
        }
#pragma unroll
        for (int k = LG - 1; k >= 0; --k) {
            const int d = 1 << k;
#pragma unroll
            for (int i = 0; i < ITEMS; i += 2 * d) {
                const float tt = pre[i] + lvl[k][i];
                pre[i + d] = tt;
                fin[i + d] = fin[i];
                fin[i] = tt;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) c[i] = fin[i];
}

// ------------------------------------------------------------------------------------------
// top-level tree over per-workgroup partials part[0..nb)
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// Every thread of the workgroup receives: root = tree sum of all partials (zero padded);
// (P, E) of the node of partial `b`.  `lds` must hold next_pow2(nb) floats when nb > 64.
// (ov_idx, ov_val): optional substitution part[ov_idx] := ov_val before the tree is built.
__device__ __forceinline__ void top_tree(const float* __restrict__ part, int nb, int b, float* lds, float& root,
                                         float& P, float& E, int ov_idx = -1, float ov_val = 0.0f) {
    const int nbp = next_pow2(nb);
    if (nbp <= 64) {
        // one wave does it in registers; every wave repeats it, so no barrier is needed
        const int lane = threadIdx.x & 63;
        float s = lane < nb ? (lane == ov_idx ? ov_val : part[lane]) : 0.0f;
        TreePath path;
        s = wave_upsweep(s, path);
        float p = 0.0f, e = s;
#pragma unroll
        for (int m = 5; m >= 0; --m) {
            const bool right = (lane >> m) & 1;
            const float t = p + (right ? path.ls[m] : path.own[m]);
            if (right) p = t; else e = t;
        }
        root = s;
        P = __shfl(p, b & 63);
        E = __shfl(e, b & 63);
        return;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbp; i += kBlock) lds[i] = i < nb ? (i == ov_idx ? ov_val : part[i]) : 0.0f;
    __syncthreads();
    for (int d = 1; d < nbp; d <<= 1) {
        for (int i = threadIdx.x; i < nbp / (2 * d); i += kBlock) {
            const int r = (i + 1) * 2 * d - 1;
            lds[r] = lds[r - d] + lds[r];
        }
        __syncthreads();
    }
    root = lds[nbp - 1];
    float p = 0.0f, e = root;
    int pos = 0;
    for (int d = nbp >> 1; d >= 1; d >>= 1) {
        const float t = p + lds[pos + d - 1];  // sum of the left child [pos, pos + d)
        if (b & d) {
            p = t;
            pos += d;
        } else {
            e = t;
        }
    }
    P = p;
    E = e;
    __syncthreads();
}

// max over per-workgroup partials (order-free)
__device__ __forceinline__ float top_max(const float* __restrict__ part, int nb, float* lds4) {
    float m = -__builtin_inff();
    for (int i = threadIdx.x; i < nb; i += kBlock) m = fmaxf(m, part[i]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
    return m;
}

__device__ __forceinline__ float block_max(float m, float* lds4) {
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}

// ------------------------------------------------------------------------------------------
// Top-level tree for up to 1024 per-workgroup partials, shaped like a tile: every thread owns four
// consecutive partials (zero beyond nb; a zero-padded tree has the same root and the same (P, E)).
// Costs one LDS exchange for the up-sweep (shareable through the *_n form) and, when a leaf's
// (P, E) is wanted, one more to broadcast it.
// ------------------------------------------------------------------------------------------
constexpr int kTopItems = 4;
constexpr int kMaxTopBlock = kBlock * kTopItems;

__device__ __forceinline__ void top_load(const float* __restrict__ part, int nb, float (&x)[kTopItems],
                                         int ov_idx = -1, float ov_val = 0.0f) {
#pragma unroll
    for (int i = 0; i < kTopItems; ++i) {
        const int e = threadIdx.x * kTopItems + i;
        x[i] = e < nb ? (e == ov_idx ? ov_val : part[e]) : 0.0f;
    }
}

__device__ __forceinline__ float top_load_max(const float* __restrict__ part, int nb) {
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < kTopItems; ++i) {
        const int e = threadIdx.x * kTopItems + i;
        if (e < nb) m = fmaxf(m, part[e]);
    }
    return m;
}

// (P, E) of leaf `b` from the up-swept top tile; `bc` = 2 floats of LDS; includes one barrier.
__device__ __forceinline__ void top_leaf(const float (&x)[kTopItems], const TreePath& path, float root, int b,
                                         float* bc, float& P, float& E) {
    float p = 0.0f, e = root;
    block_descend(p, e, path);
    if ((int)threadIdx.x == (b >> 2)) {
        // two in-chunk levels
        const float s01 = x[0] + x[1];
        float t = p + s01;
        if (b & 2) p = t; else e = t;
        t = p + ((b & 2) ? x[2] : x[0]);
        if (b & 1) p = t; else e = t;
        bc[0] = p;
        bc[1] = e;
    }
    __syncthreads();
    P = bc[0];
    E = bc[1];
}

// ------------------------------------------------------------------------------------------
// Two-level logsumexp (specified in include/fbsmi_math.h): a workgroup tile publishes
// (m_t = max, s_t = tree-sum exp(x - m_t')), consumers combine the per-tile pairs.  One grid-wide
// dependency instead of the two of "global max, then sum".
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float finite_or_zero_f(float m) { return (fabsf(m) <= 3.40282347e+38f) ? m : 0.0f; }

// producer: l[] = this thread's log-weights (-inf where there is no element).  lds_a, lds_b: 4
// floats each, untouched since the last barrier.
template <int ITEMS>
__device__ __forceinline__ void block_lse_partial(const float (&l)[ITEMS], float* lds_a, float* lds_b, float& m_out,
                                                  float& s_out) {
    float m = l[0];
#pragma unroll
    for (int i = 1; i < ITEMS; ++i) m = fmaxf(m, l[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) lds_a[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(lds_a[0], lds_a[1]), fmaxf(lds_a[2], lds_a[3]));
    const float mp = finite_or_zero_f(m);
    float x[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) x[i] = fbsmi_expf(l[i] - mp);
    float sv[1] = {chunk_total<ITEMS>(x)}, tot[1];
    TreePath path[1];
    block_upsweep_n<1>(sv, path, lds_b, tot);
    m_out = m;
    s_out = tot[0];
}

// two tiles at once (a workgroup that owns slot t of tile A and slot t of tile B): one LDS exchange per stage
__device__ __forceinline__ void block_lse_partial2(float lA, float lB, float* lds8a, float* lds8b, float& mA, float& sA,
                                                   float& mB, float& sB) {
    float a = wave_max(lA), b = wave_max(lB);
    if ((threadIdx.x & 63) == 0) {
        lds8a[threadIdx.x >> 6] = a;
        lds8a[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = fmaxf(fmaxf(lds8a[0], lds8a[1]), fmaxf(lds8a[2], lds8a[3]));
    b = fmaxf(fmaxf(lds8a[4], lds8a[5]), fmaxf(lds8a[6], lds8a[7]));
    float sv[2] = {fbsmi_expf(lA - finite_or_zero_f(a)), fbsmi_expf(lB - finite_or_zero_f(b))}, tot[2];
    TreePath path[2];
    block_upsweep_n<2>(sv, path, lds8b, tot);
    mA = a;
    sA = tot[0];
    mB = b;
    sB = tot[1];
}

// consumer (up to kMaxTopBlock tiles): lse and the raw global max
__device__ __forceinline__ void lse_from_partials(const float* __restrict__ pmax, const float* __restrict__ psum, int nb,
                                                  float* lds_a, float* lds_b, float& lse, float& Mraw) {
    if (nb <= kBlock) {   // one tile per thread: the same canonical tree (zero leaves add exactly), a quarter of the exps
        const int e = threadIdx.x;
        const float m1 = e < nb ? pmax[e] : -__builtin_inff();
        const float s1 = e < nb ? psum[e] : 0.0f;
        float m = wave_max(m1);
        if ((threadIdx.x & 63) == 0) lds_a[threadIdx.x >> 6] = m;
        __syncthreads();
        Mraw = fmaxf(fmaxf(lds_a[0], lds_a[1]), fmaxf(lds_a[2], lds_a[3]));
        const float Mp = finite_or_zero_f(Mraw);
        float sv[1] = {e < nb ? s1 * fbsmi_expf(finite_or_zero_f(m1) - Mp) : 0.0f}, tot[1];
        TreePath path[1];
        block_upsweep_n<1>(sv, path, lds_b, tot);
        lse = fbsmi_logf(tot[0]) + Mp;
        return;
    }
    float m4[kTopItems], s4[kTopItems];
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < kTopItems; ++i) {
        const int e = threadIdx.x * kTopItems + i;
        m4[i] = e < nb ? pmax[e] : -__builtin_inff();
        s4[i] = e < nb ? psum[e] : 0.0f;
        m = fmaxf(m, m4[i]);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) lds_a[threadIdx.x >> 6] = m;
    __syncthreads();
    Mraw = fmaxf(fmaxf(lds_a[0], lds_a[1]), fmaxf(lds_a[2], lds_a[3]));
    const float Mp = finite_or_zero_f(Mraw);
    float t4[kTopItems];
#pragma unroll
    for (int i = 0; i < kTopItems; ++i) {
        const int e = threadIdx.x * kTopItems + i;
        t4[i] = e < nb ? s4[i] * fbsmi_expf(finite_or_zero_f(m4[i]) - Mp) : 0.0f;
    }
    float sv[1] = {chunk_total<kTopItems>(t4)}, tot[1];
    TreePath path[1];
    block_upsweep_n<1>(sv, path, lds_b, tot);
    lse = fbsmi_logf(tot[0]) + Mp;
}

// ------------------------------------------------------------------------------------------
// Fast exact bisection.  The first Lh <= 8 levels of the fixed bisection visit a fixed implicit
// tree of at most 255 array positions: a workgroup gathers them into LDS once (one memory round
// trip), and every search walks them there.  The remaining levels go three at a time: the 7
// candidate positions of a 3-level subtree are loaded together, so a search costs
// ceil(rem/3) dependent round trips instead of rem.  Running a few levels more than
// ceil(log2(n+1)) is harmless: once the interval has collapsed every further level repeats a
// comparison already made (same operands, same outcome).
// ------------------------------------------------------------------------------------------
constexpr int kHeapLevels = 8;
constexpr int kHeapSize = 1 << kHeapLevels;

__device__ __forceinline__ int heap_node_mid(int t, int n) {
    int lo = 0, hi = n;
    const int l = 31 - __builtin_clz(t);
    for (int i = l - 1; i >= 0; --i) {
        const int mid = (lo + hi) >> 1;
        if ((t >> i) & 1) lo = mid; else hi = mid;
    }
    return (lo + hi) >> 1;
}

__device__ __forceinline__ int bisect_heap(const float* __restrict__ a, int n, int levels, const float* heap,
                                           float q) {
    int lo = 0, hi = n, t = 1;
    const int Lh = levels < kHeapLevels ? levels : kHeapLevels;
    for (int l = 0; l < Lh; ++l) {
        const int mid = (lo + hi) >> 1;
        const bool gl = q <= heap[t];
        hi = gl ? mid : hi;
        lo = gl ? lo : mid;
        t = 2 * t + (gl ? 0 : 1);
    }
    for (int rem = levels - Lh; rem > 0; rem -= 3) {
        const int m1 = (lo + hi) >> 1;
        const int m2l = (lo + m1) >> 1, m2r = (m1 + hi) >> 1;
        const int m3a = (lo + m2l) >> 1, m3b = (m2l + m1) >> 1, m3c = (m1 + m2r) >> 1, m3d = (m2r + hi) >> 1;
        const float v1 = a[m1], v2l = a[m2l], v2r = a[m2r], v3a = a[m3a], v3b = a[m3b], v3c = a[m3c], v3d = a[m3d];
        const bool g1 = q <= v1;
        hi = g1 ? m1 : hi;
        lo = g1 ? lo : m1;
        const int m2 = g1 ? m2l : m2r;
        const bool g2 = q <= (g1 ? v2l : v2r);
        hi = g2 ? m2 : hi;
        lo = g2 ? lo : m2;
        const int m3 = g1 ? (g2 ? m3a : m3b) : (g2 ? m3c : m3d);
        const bool g3 = q <= (g1 ? (g2 ? v3a : v3b) : (g2 ? v3c : v3d));
        hi = g3 ? m3 : hi;
        lo = g3 ? lo : m3;
    }
    return hi;
}

// Building blocks of the one-slot-per-thread kernels.
// The LDS part of the walk over a heap of Lh levels (node 0 unused):
__device__ __forceinline__ void bisect_lds_levels(int n, int Lh, const float* heap, float q, int& lo, int& hi) {
    lo = 0;
    hi = n;
    int t = 1;
    for (int l = 0; l < Lh; ++l) {
        const int mid = (lo + hi) >> 1;
        const bool gl = q <= heap[t];
        hi = gl ? mid : hi;
        lo = gl ? lo : mid;
        t = 2 * t + (gl ? 0 : 1);
    }
}

// Two independent searches walking the LDS levels in lockstep (their LDS reads overlap), and taking a
// three-level round together (all fourteen probes issued before the first is used)
__device__ __forceinline__ void bisect_lds_levels_x2(int n, int Lh, const float* heap, const float (&q)[2], int (&lo)[2],
                                                     int (&hi)[2]) {
    int l0 = 0, h0 = n, t0 = 1, l1 = 0, h1 = n, t1 = 1;
    for (int l = 0; l < Lh; ++l) {
        const float v0 = heap[t0], v1 = heap[t1];
        const int m0 = (l0 + h0) >> 1, m1 = (l1 + h1) >> 1;
        const bool g0 = q[0] <= v0, g1 = q[1] <= v1;
        h0 = g0 ? m0 : h0;
        l0 = g0 ? l0 : m0;
        t0 = 2 * t0 + (g0 ? 0 : 1);
        h1 = g1 ? m1 : h1;
        l1 = g1 ? l1 : m1;
        t1 = 2 * t1 + (g1 ? 0 : 1);
    }
    lo[0] = l0; hi[0] = h0; lo[1] = l1; hi[1] = h1;
}

__device__ __forceinline__ void bisect_round3_x2(const float* __restrict__ a, int (&lo)[2], int (&hi)[2], const float (&q)[2],
                                                 const bool (&on)[2]) {
    int m1[2], m2l[2], m2r[2], m3a[2], m3b[2], m3c[2], m3d[2];
    float v1[2], v2l[2], v2r[2], v3a[2], v3b[2], v3c[2], v3d[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int l0 = lo[k], h0 = hi[k];
        m1[k] = (l0 + h0) >> 1;
        m2l[k] = (l0 + m1[k]) >> 1;
        m2r[k] = (m1[k] + h0) >> 1;
        m3a[k] = (l0 + m2l[k]) >> 1;
        m3b[k] = (m2l[k] + m1[k]) >> 1;
        m3c[k] = (m1[k] + m2r[k]) >> 1;
        m3d[k] = (m2r[k] + h0) >> 1;
        v1[k] = v2l[k] = v2r[k] = v3a[k] = v3b[k] = v3c[k] = v3d[k] = 0.0f;
        if (on[k]) {
            v1[k] = a[m1[k]]; v2l[k] = a[m2l[k]]; v2r[k] = a[m2r[k]];
            v3a[k] = a[m3a[k]]; v3b[k] = a[m3b[k]]; v3c[k] = a[m3c[k]]; v3d[k] = a[m3d[k]];
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!on[k]) continue;
        int l = lo[k], h = hi[k];
        const bool g1 = q[k] <= v1[k];
        h = g1 ? m1[k] : h;
        l = g1 ? l : m1[k];
        const int m2 = g1 ? m2l[k] : m2r[k];
        const bool g2 = q[k] <= (g1 ? v2l[k] : v2r[k]);
        h = g2 ? m2 : h;
        l = g2 ? l : m2;
        const int m3 = g1 ? (g2 ? m3a[k] : m3b[k]) : (g2 ? m3c[k] : m3d[k]);
        const bool g3 = q[k] <= (g1 ? (g2 ? v3a[k] : v3b[k]) : (g2 ? v3c[k] : v3d[k]));
        h = g3 ? m3 : h;
        l = g3 ? l : m3;
        lo[k] = l;
        hi[k] = h;
    }
}

// K independent searches taking a three-level round together: all 7 K probes are issued before the first is used
// (`on[k]` = false: no loads for search k, its interval untouched)
template <int K>
__device__ __forceinline__ void bisect_round3_xn(const float* __restrict__ a, int (&lo)[K], int (&hi)[K], const float (&q)[K],
                                                 const bool (&on)[K]) {
    int m1[K], m2l[K], m2r[K], m3a[K], m3b[K], m3c[K], m3d[K];
    float v1[K], v2l[K], v2r[K], v3a[K], v3b[K], v3c[K], v3d[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int l0 = lo[k], h0 = hi[k];
        m1[k] = (l0 + h0) >> 1;
        m2l[k] = (l0 + m1[k]) >> 1;
        m2r[k] = (m1[k] + h0) >> 1;
        m3a[k] = (l0 + m2l[k]) >> 1;
        m3b[k] = (m2l[k] + m1[k]) >> 1;
        m3c[k] = (m1[k] + m2r[k]) >> 1;
        m3d[k] = (m2r[k] + h0) >> 1;
        // no load sits under a divergent branch (the compiler would drain the whole memory queue where the paths meet):
        // searches that are off read element 0 -- one broadcast line -- and ignore it
        const bool o = on[k];
        v1[k] = a[o ? m1[k] : 0]; v2l[k] = a[o ? m2l[k] : 0]; v2r[k] = a[o ? m2r[k] : 0];
        v3a[k] = a[o ? m3a[k] : 0]; v3b[k] = a[o ? m3b[k] : 0]; v3c[k] = a[o ? m3c[k] : 0]; v3d[k] = a[o ? m3d[k] : 0];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int l = lo[k], h = hi[k];
        const bool g1 = q[k] <= v1[k];
        h = g1 ? m1[k] : h;
        l = g1 ? l : m1[k];
        const int m2 = g1 ? m2l[k] : m2r[k];
        const bool g2 = q[k] <= (g1 ? v2l[k] : v2r[k]);
        h = g2 ? m2 : h;
        l = g2 ? l : m2;
        const int m3 = g1 ? (g2 ? m3a[k] : m3b[k]) : (g2 ? m3c[k] : m3d[k]);
        const bool g3 = q[k] <= (g1 ? (g2 ? v3a[k] : v3b[k]) : (g2 ? v3c[k] : v3d[k]));
        h = g3 ? m3 : h;
        l = g3 ? l : m3;
        lo[k] = on[k] ? l : lo[k];
        hi[k] = on[k] ? h : hi[k];
    }
}

// Three levels in one memory round trip (`on` = false: no loads, interval untouched)
__device__ __forceinline__ void bisect_round3(const float* __restrict__ a, int& lo, int& hi, float q, bool on) {
    if (!on) return;
    const int l0 = lo, h0 = hi;
    const int m1 = (l0 + h0) >> 1;
    const int m2l = (l0 + m1) >> 1, m2r = (m1 + h0) >> 1;
    const int m3a = (l0 + m2l) >> 1, m3b = (m2l + m1) >> 1, m3c = (m1 + m2r) >> 1, m3d = (m2r + h0) >> 1;
    const float v1 = a[m1], v2l = a[m2l], v2r = a[m2r], v3a = a[m3a], v3b = a[m3b], v3c = a[m3c], v3d = a[m3d];
    int l = l0, h = h0;
    const bool g1 = q <= v1;
    h = g1 ? m1 : h;
    l = g1 ? l : m1;
    const int m2 = g1 ? m2l : m2r;
    const bool g2 = q <= (g1 ? v2l : v2r);
    h = g2 ? m2 : h;
    l = g2 ? l : m2;
    const int m3 = g1 ? (g2 ? m3a : m3b) : (g2 ? m3c : m3d);
    const bool g3 = q <= (g1 ? (g2 ? v3a : v3b) : (g2 ? v3c : v3d));
    h = g3 ? m3 : h;
    l = g3 ? l : m3;
    lo = l;
    hi = h;
}

// A search whose query is the same for the whole workgroup (the rotation J): after the LDS levels
// the interval [lo, hi) is workgroup-uniform; once it is at most kBlock wide the workgroup fetches
// it whole in ONE round trip and every thread finishes the walk in LDS.  win: kBlock floats.
__device__ __forceinline__ int bisect_uniform(const float* __restrict__ a, int n, int levels, int Lh, const float* heap,
                                              float* win, float q) {
    int lo, hi;
    bisect_lds_levels(n, Lh, heap, q, lo, hi);
    int rem = levels - Lh;
    while (rem > 0 && hi - lo > kBlock) {
        bisect_round3(a, lo, hi, q, true);
        rem -= 3;
    }
    if (rem > 0) {   // uniform branch: q, lo, hi and rem are the same in every thread
        const int w0 = lo, e = lo + (int)threadIdx.x;
        win[threadIdx.x] = a[e < n ? e : n - 1];
        __syncthreads();
        for (int l = 0; l < rem; ++l) {
            const int mid = (lo + hi) >> 1;
            const bool gl = q <= win[mid - w0];
            hi = gl ? mid : hi;
            lo = gl ? lo : mid;
        }
    }
    return hi;
}

// logsumexp's "amax if finite else 0" (jax.scipy.special.logsumexp)
__device__ __forceinline__ float finite_or_zero(float m) { return (fabsf(m) <= 3.40282347e+38f) ? m : 0.0f; }

}  // namespace fbsmi
