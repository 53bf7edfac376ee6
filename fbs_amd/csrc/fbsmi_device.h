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
    const uint64_t half = (n + 1) >> 1;
    uint32_t o0, o1;
    if (i < half) {
        const uint64_t j = i + half;
        threefry2x32(k0, k1, (uint32_t)i, j < n ? (uint32_t)j : 0u, o0, o1);
        return o0;
    }
    threefry2x32(k0, k1, (uint32_t)(i - half), (uint32_t)i, o0, o1);
    return o1;
}

__host__ __device__ __forceinline__ float uniform_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i) {
    return fbsmi_bits_to_unit(random_bits_at(k0, k1, n, i));
}

__host__ __device__ __forceinline__ float normal_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t i) {
    return fbsmi_bits_to_normal(random_bits_at(k0, k1, n, i));
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
};

// Up-sweep. `s` is the tree sum of this thread's own chunk.  Returns the tile total to every
// thread.  `lds4` is 4 floats of LDS scratch (reusable after the call returns).
__device__ __forceinline__ float block_upsweep(float s, TreePath& path, float* lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const float o = __shfl_xor(s, 1 << m);
        path.ls[m] = o;
        path.own[m] = s;
        s = ((lane >> m) & 1) ? o + s : s + o;
    }
    __syncthreads();  // protect lds4 against a previous use
    if (lane == 0) lds4[wave] = s;
    __syncthreads();
    const float w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
    const float s01 = w0 + w1, s23 = w2 + w3;
    path.own[6] = s;
    path.ls[6] = (wave & 2) ? ((wave & 1) ? w2 : w3) : ((wave & 1) ? w0 : w1);
    path.ls[7] = s01;
    return s01 + s23;
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
        float ls[6], own[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const float o = __shfl_xor(s, 1 << m);
            ls[m] = o;
            own[m] = s;
            s = ((lane >> m) & 1) ? o + s : s + o;
        }
        float p = 0.0f, e = s;
#pragma unroll
        for (int m = 5; m >= 0; --m) {
            const bool right = (lane >> m) & 1;
            const float t = p + (right ? ls[m] : own[m]);
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
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
    return m;
}

__device__ __forceinline__ float block_max(float m, float* lds4) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}

// logsumexp's "amax if finite else 0" (jax.scipy.special.logsumexp)
__device__ __forceinline__ float finite_or_zero(float m) { return (fabsf(m) <= 3.40282347e+38f) ? m : 0.0f; }

}  // namespace fbsmi
