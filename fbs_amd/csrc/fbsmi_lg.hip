// fbsmi_lg.hip -- the fused linear-Gaussian Gibbs sweep: gibbs_kernel (fbs/samplers/gibbs.py:68-168)
// -> csmc.forward_pass (fbs/samplers/csmc/csmc.py:80-164) with conditional killing resampling
// (fbs/samplers/csmc/resamplings.py:40-88), force_move (gibbs.py:171-214), the forward noising
// sampler (fbs/sdes/linear.py:190-221) and the three model closures of
// experiments/toy/gp_gibbs.py:120-135 folded in, entirely on the device.
//
// The general SMC step is three kernels (kernel boundaries are the cheapest grid-wide synchronisation on
// MI355X, ~1.7 us):
//
//   norm  : lse from the (max, sumexp) pairs the previous kernel published per workgroup (two-level
//           logsumexp, include/fbsmi_math.h); w = exp(lw - lse);
//           w_max = exp(max lw - lse) (fbsmi_expf is monotone, so this IS max_i w_i);
//           per-workgroup tree sums of w and of J_prob                        [csmc.py:146,139]
//   cdf   : canonical-tree cumsum of w and of J_prob (J_prob[i*] needs the total first)
//                                                                        [resamplings.py:74-84]
//   prop  : J ~ Cat(J_prob); per slot: rotate by j*-J, kill test, Cat(w) draw for killed slots,
//           pin, gather the ancestor, Euler-Maruyama step, pin the reference, Gaussian
//           log-weight, per-workgroup (max, sumexp)                      [resamplings.py:71-86, csmc.py:140-145]
//
// Particle state is structure-of-arrays u[r][p] so that every per-slot access is coalesced.
// A whole sweep (2T + ~10 launches when N is a power of two, 3T + ~10 otherwise) is captured once into a hipGraph and replayed.
//
// Variants of the step, chosen at handle creation (all bit-identical to the oracle):
//   N <= 256 (one logsumexp tile, one workgroup)   : lgw_pre_body does norm + cdf + searches in LDS;
//       narrow models run the whole T loop in ONE launch (k_lg_sweep1), particle filters too (k_filt_sweep1)
//   wide models, 16 < max(du, dv) <= 128           : row-major particles, drift on the f32 matrix cores
//       (k_lgw_gemm: v_mfma_f32_16x16x4_f32 == ascending fmaf chain); N <= 256: one launch per step with the
//       prologue fused in (noise and resampler uniforms drawn a step ahead by blocks of the previous launch); N > 256:
//       norm -> cdf -> k_lgw_anc -> k_lgw_gemm -> k_lgw_lse; from ~700 tiled workgroups per launch k_lgw_gemm_fat (one
//       workgroup per slot tile walks all row tiles; the three small launches carry extra blocks that draw the step's
//       noise; the log-density terms are summed in the kernel, through LDS) -> k_lg_lwpart
//   N > 131072                                      : 4 / 16 slots per thread, four launches per step: norm -> cdf -> k_lg_heaps
//       (compact bisection heaps) -> k_lg_propQ (lane-major slots, kill tests first, the killed sources' searches compacted
//       through an LDS queue); chunks of a thread move as 16-byte accesses.  (The draws by a launch of their own, one
//       Threefry call per element pair, through HBM: measured 10 % slower at 2^22, 6 % at 2^20 -- not kept.)
//   N a power of two, 512 .. 65536                  : TWO kernels per step -- the bisection over the canonical cumsum is a
//       descent of the summation tree, so norm publishes tree nodes and k_lg_prop1t / k_lg_prop2t walk them; no cdf
//   N = 2^k + 1 (explicit_final on such an ensemble) : the same two kernels over the first 2^k slots' tree + one extra tile
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fbsmi.h"
#include "fbsmi_device.h"
#include "fbsmi_host.h"

namespace fbsmi {

constexpr int kMaxNbSweep = 1024;  // workgroup partials one in-kernel top tree handles
constexpr int kNumProfKernels = 3; // norm, cdf, prop

struct LgDev {
    int C;           // independent chains batched in every launch (blockIdx.y): jax.vmap over chains
    int Ctot, c0;    // this handle drives chains c0 .. c0 + C - 1 of a batch of Ctot (fbsmi_lg_sweep_set_group; default C, 0)
    int plus1;       // N = 2^k + 1 (explicit_final on a power-of-two ensemble): the two-launch step over the first 2^k slots'
                     // summation tree plus one extra tile that holds the last slot (see tree_build)
    int pin;         // small ensembles: the two-launch step's grids are 8x as wide and only every eighth block works, so the
                     // whole step runs on ONE XCD (blocks b and b + 8 share one) and its hand-offs stay in that XCD's L2
    int nz;          // large wide ensembles: the norm / cdf / ancestor launches of a step (a few dozen workgroups each) carry this
                     // many EXTRA blocks, which draw a third each of the step's noise for the drift kernel (wide_noise_share)
    int N;           // rows of the particle system (nparticles, +1 when explicit_final)
    int nparticles;
    int du, dv, D, T;
    int nb;          // workgroups of every tile kernel
    int levels;      // bisection levels for N
    int eb, ef, store;
    float dt;
    float lw_init;   // -log(nparticles)
    const float *G, *g, *sd, *lognorm, *F, *sqQ;
    // per-sweep inputs (internal copies)
    uint32_t* key;   // [2]     master key of the chain driver (shared)
    uint32_t* keys;  // [C][2]  per-chain sweep keys
    float* x0;       // [C][du]
    float* y0;       // [dv]    shared by all chains (vmap in_axes=None, gp_gibbs.py:173)
    int32_t* bs;     // [C][T+1]
    // derived per sweep
    uint32_t* keytab;  // [T][8]: key_1, key_2, key_3 of the killing resampler, key_transition
    uint32_t* misc;    // [16]: 0 key_fwd, 2 key_init, 4 key_x0(force_move), 6 key_us, 8 key_bs, 10 key_bwd
    float* xi1;        // [T][D] noise of the first forward path
    float* xi2;        // [T][D] noise of the second forward path (us_star_next)
    float* path;       // [T+1][D]
    float* us_star;    // [T+1][du]
    float* vs;         // [T+1][dv]
    // state
    float* u0;         // [du][N]
    float* u1;
    float* lw;         // [N] unnormalised log-weights
    float* lwn;        // [N] normalised log-weights of the last normalise
    float* w;          // [N]
    float* cdf;        // [N]
    float* cdfJ;       // [N]
    float *bmax, *bsumexp, *bsumw, *bsumJ;  // [nb]
    float *hpW, *hpJ;                       // compact bisection heaps of cdf / cdfJ (one slot per thread only)
    const int32_t* hp_map;                  // [N] heap node (low 16 bits) and depth (high) an element is the midpoint of, or 0
    int wide;                               // 16 < max(du, dv) <= 128: row-major particles u0/u1 [N][du], MFMA drift
    float* lpw;                             // [N][dv rounded up to 4] per-row log-density terms of the wide path
    int32_t* anc;                           // [N] ancestors of the current step (wide path)
    float* xiw;                             // [2][N][du] a step's noise, drawn ahead of the launch that uses it (wide models): slot s & 1
                                            // for the one-tile Gibbs step (drawn DURING the previous launch), slot 0 for k_lgw_noise
    float* uw;                              // [2][2][N] one-tile wide Gibbs step: the kill-test and redraw uniforms of every source slot
                                            // (resamplings.py:71-74), drawn ahead like the noise: slot s & 1, u1 then u2 (nullable)
    int lh_w, lh_j;                         // their depths
    // two-launch step (N a power of two, 2..256 tiles): the searches walk the summation tree itself, so no cdf is written
    float2* trW;                            // [nb][64]: per tile, heap-ordered nodes (sum of the node's left half, w at its midpoint)
    float2* trWtop;                         // [nb][kMidN]: the first kMidLv levels of every tile's heap, compact (LDS staging)
    float* wfirst;                          // [nb]: w of every tile's first element
    float* scal;       // [16]: 0 lse, 1 w_max
    int32_t* As;       // [T][N] or null
    float* uss;        // [T+1][N][du] or null
    float* lwss;       // [T+1][N] or null
    float* usT;        // [N][du] row-major copy of the final particles
    // outputs (internal)
    float* x0n;        // [du]
    float* usn;        // [T+1][du]
    int32_t* bsn;      // [T+1]
    uint8_t* acc;      // [T+1]
    // chain bookkeeping
    float** x0s_slot;  // device slot holding the x0s pointer (or null)
    int32_t* counter;  // device sweep counter
    unsigned long long* dbg;  // [64] in-kernel stamps (diagnostic build -DFBSMI_STAMPS only)
    // fused particle filters (bootstrap_filter / pmcmc_filter_step)
    int flow;        // 0 bootstrap_filter (smc.py:58-74), 1 pmcmc_filter_step (smc.py:138-152)
    int systematic;  // resampling: 0 stratified, 1 systematic (resampling.py:43-59)
    float logn;      // log(nparticles)
    float* ell;      // [C] running log-likelihood (flow 1) / negative log-likelihood (flow 0)
};

#ifdef FBSMI_STAMPS
// diagnostic build: one lane of the middle workgroup records (100 MHz wall clock, shader clock)
#define FBSMI_STAMP(i)                                                                   \
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0) {            \
        d.dbg[2 * (i)] = __builtin_amdgcn_s_memrealtime();                               \
        d.dbg[2 * (i) + 1] = __builtin_amdgcn_s_memtime();                               \
    }
// ... and every workgroup of the LAST step of the LAST sweep of a chain call records its entry / exit time in a slot of its
// own (in the cdfJ array, which the two-launch step does not use; view 8): how far a launch's first and last workgroup are
// apart.  (A shared minimum / maximum would serialise 256 atomics at the end of the launch and measure itself.)
#define FBSMI_SPAN_IN(i, step, last_sweep)                                                                   \
    if (threadIdx.x == 0 && (step) == d.T - 1 && *d.counter == (last_sweep))                                  \
        reinterpret_cast<unsigned long long*>(d.cdfJ)[(i) * 1024 + 2 * (blockIdx.x + gridDim.x * blockIdx.y)] = __builtin_amdgcn_s_memrealtime();
#define FBSMI_SPAN_OUT(i, step, last_sweep)                                                                  \
    if (threadIdx.x == 0 && (step) == d.T - 1 && *d.counter == (last_sweep))                                  \
        reinterpret_cast<unsigned long long*>(d.cdfJ)[(i) * 1024 + 2 * (blockIdx.x + gridDim.x * blockIdx.y) + 1] = __builtin_amdgcn_s_memrealtime();
#else
#define FBSMI_STAMP(i)
#define FBSMI_SPAN_IN(i, step, last_sweep)
#define FBSMI_SPAN_OUT(i, step, last_sweep)
#endif

// The view of chain c: every per-chain array advanced to that chain's slice (all per-chain arrays
// are laid out [C][...]).
// depths of the compact heaps the cdf kernel publishes for k_lg_prop1
constexpr int kHeapLevelsW = 11, kHeapSizeW = 1 << kHeapLevelsW;
constexpr int kHeapLevelsJ = 8, kHeapSizeJ = 1 << kHeapLevelsJ;
constexpr int kMidLv = 3, kMidN = 1 << kMidLv;   // levels (nodes) of every tile's tree the Euler kernel keeps in LDS (4 measured: staging 32 KB per workgroup costs more than the two probes it saves)
constexpr int kTreeNodes = 64;   // nodes of a tile's summation tree that are published (down to blocks of 8 leaves)

__device__ __forceinline__ LgDev chain_view(LgDev d, int c) {
    const size_t N = d.N, T = d.T, D = d.D, du = d.du, dv = d.dv, nb = d.nb;
    d.keys += 2 * (size_t)c;
    d.x0 += du * c;
    d.bs += (T + 1) * c;
    d.keytab += 8 * T * c;
    d.misc += 16 * (size_t)c;
    d.xi1 += T * D * c;
    d.xi2 += T * D * c;
    d.us_star += (T + 1) * du * c;
    d.vs += (T + 1) * dv * c;
    d.u0 += N * du * c;
    d.u1 += N * du * c;
    d.lw += N * c;
    d.lwn += N * c;
    d.w += N * c;
    d.cdf += N * c;
    d.cdfJ += N * c;
    d.bmax += nb * c;
    d.bsumexp += nb * c;
    d.bsumw += nb * c;
    d.bsumJ += nb * c;
    if (d.lpw) {
        d.lpw += (size_t)((d.dv + 3) & ~3) * N * c;
        d.anc += N * c;
        d.xiw += 2 * N * du * c;
        if (d.uw) d.uw += 4 * N * c;
    }
    if (d.hpW) {
        d.hpW += (size_t)kHeapSizeW * c;
        d.hpJ += (size_t)kHeapSizeJ * c;
    }
    if (d.trW) {
        d.trW += kTreeNodes * nb * c;
        d.trWtop += kMidN * nb * c;
        d.wfirst += nb * c;
    }
    d.scal += 16 * (size_t)c;
    if (d.As) d.As += T * N * c;
    if (d.uss) d.uss += (T + 1) * N * du * c;
    if (d.lwss) d.lwss += (T + 1) * N * c;
    d.usT += N * du * c;
    d.x0n += du * c;
    d.usn += (T + 1) * du * c;
    d.bsn += (T + 1) * c;
    d.acc += (T + 1) * c;
    if (d.ell) d.ell += c;
    return d;
}

// element (row r, slot p) of a particle buffer: structure-of-arrays [du][N] for narrow models (every
// per-slot access coalesced), row-major [N][du] for wide ones (the ancestor gather moves whole rows)
__device__ __forceinline__ size_t u_at(const LgDev& d, int r, int p) {
    return d.wide ? (size_t)p * d.du + r : (size_t)r * d.N + p;
}

// ------------------------------------------------------------------------------------------
// sweep prologue: key derivation, forward noising path
// ------------------------------------------------------------------------------------------
// gibbs.py:126,147 ; csmc.py:65,150,157,136 ; resamplings.py:66
__global__ void __launch_bounds__(kBlock) k_lg_keys(LgDev dd, int chain) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ uint32_t sk[4];
    if (threadIdx.x == 0) {
        uint32_t k0 = d.keys[0], k1 = d.keys[1];
        if (chain) {
            // chain driver: key, subkey = split(key) (the master key itself is advanced by
            // k_lg_advance); one chain sweeps with subkey (tests/test_gibbs.py:116), a batch of C
            // chains with split(subkey, C)[c] (experiments/toy/gp_gibbs.py:183-185)
            uint32_t b0, b1;
            split_at(d.key[0], d.key[1], 2, 1, b0, b1);
            if (d.Ctot > 1) split_at(b0, b1, d.Ctot, d.c0 + blockIdx.y, k0, k1);
            else { k0 = b0; k1 = b1; }
        }
        uint32_t f0, f1, c0, c1;
        split_at(k0, k1, 3, 0, f0, f1);  // key_fwd
        split_at(k0, k1, 3, 1, c0, c1);  // key_csmc  (key_bridge unused: marg_y=False)
        uint32_t cf0, cf1;
        if (d.eb) {
            split_at(c0, c1, 4, 0, cf0, cf1);                    // key_csmc_fwd
            split_at(c0, c1, 4, 1, d.misc[4], d.misc[5]);        // key_csmc_x0
            split_at(c0, c1, 4, 2, d.misc[6], d.misc[7]);        // key_csmc_bwd_us
            split_at(c0, c1, 4, 3, d.misc[8], d.misc[9]);        // key_csmc_bwd_bs
        } else {
            split_at(c0, c1, 2, 0, cf0, cf1);                    // key_fwd of csmc_kernel
            split_at(c0, c1, 2, 1, d.misc[10], d.misc[11]);      // key_bwd
            d.misc[6] = 0; d.misc[7] = 0;
        }
        d.misc[0] = f0;
        d.misc[1] = f1;
        split_at(cf0, cf1, 2, 0, d.misc[2], d.misc[3]);          // key_init
        split_at(cf0, cf1, 2, 1, sk[0], sk[1]);                  // key_scan
    }
    __syncthreads();
    const uint32_t s0 = sk[0], s1 = sk[1];
    for (int s = threadIdx.x; s < d.T; s += kBlock) {
        uint32_t q0, q1, r0, r1;
        split_at(s0, s1, d.T, s, q0, q1);  // keys[s]
        uint32_t* kt = d.keytab + 8 * s;
        split_at(q0, q1, 2, 0, r0, r1);    // key_resampling
        split_at(q0, q1, 2, 1, kt[6], kt[7]);  // key_transition
        split_at(r0, r1, 3, 0, kt[0], kt[1]);
        split_at(r0, r1, 3, 1, kt[2], kt[3]);
        // key_3 draws the single uniform of the rotation J (resamplings.py:84): draw it here, once per sweep,
        // instead of once per thread of every step kernel
        uint32_t c0, c1;
        split_at(r0, r1, 3, 2, c0, c1);
        kt[4] = __float_as_uint(uniform_at(c0, c1, 1, 0));
        kt[5] = 0;
    }
}

// normal(key, (T, D)) for both forward paths (linear.py:220)
__global__ void k_lg_noise(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const uint64_t n = (uint64_t)d.T * d.D;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        d.xi1[i] = normal_at(d.misc[0], d.misc[1], n, i);
        if (d.eb) d.xi2[i] = normal_at(d.misc[6], d.misc[7], n, i);
    }
}

// x_{k+1} = F_k x_k + sqrt(Q_k) xi_k (linear.py:211-221), one thread per coordinate; then the
// time reversal and unpack of gibbs.py:128-130.  which = 0: from (x0, y0) -> us_star, vs;
// which = 1: from (x0n, y0) -> usn (gibbs.py:155), bs_next, acc (gibbs.py:156,168).
// The recurrence is sequential in k; its operands are not, so they are fetched kPathChunk steps
// ahead (independent loads in flight) while the previous chunk is being folded in.
constexpr int kPathChunk = 16;

__global__ void k_lg_path(LgDev dd, int which) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const float* xi = which ? d.xi2 : d.xi1;
    if (c < d.D) {
        float x = c < d.du ? (which ? d.x0n[c] : d.x0[c]) : d.y0[c - d.du];
        float* dst = c < d.du ? (which ? d.usn : d.us_star) : (which ? nullptr : d.vs);
        const int stride = c < d.du ? d.du : d.dv;
        const int col = c < d.du ? c : c - d.du;
        float F[kPathChunk], S[kPathChunk], Z[kPathChunk];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int j = 0; j < kPathChunk; ++j) {
                const int k = k0 + j < d.T ? k0 + j : d.T - 1;
                F[j] = d.F[k];
                S[j] = d.sqQ[k];
                Z[j] = xi[(size_t)k * d.D + c];
            }
        };
        if (dst) dst[(size_t)d.T * stride + col] = x;
        fetch(0);
        for (int k0 = 0; k0 < d.T; k0 += kPathChunk) {
            float f[kPathChunk], sq[kPathChunk], z[kPathChunk];
#pragma unroll
            for (int j = 0; j < kPathChunk; ++j) {
                f[j] = F[j];
                sq[j] = S[j];
                z[j] = Z[j];
            }
            if (k0 + kPathChunk < d.T) fetch(k0 + kPathChunk);
#pragma unroll
            for (int j = 0; j < kPathChunk; ++j) {
                const int k = k0 + j;
                if (k < d.T) {
                    x = f[j] * x + sq[j] * z[j];
                    if (dst) dst[(size_t)(d.T - 1 - k) * stride + col] = x;
                }
            }
        }
    }
    if (which) {
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= d.T; k += gridDim.x * blockDim.x) {
            const int32_t b = randint_at(d.misc[8], d.misc[9], (uint64_t)d.T + 1, (uint64_t)k, 0, d.nparticles);
            d.acc[k] = b != d.bs[k];
            d.bsn[k] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------
// model closures
// ------------------------------------------------------------------------------------------
template <int DMAX>
struct StepTables {
    const float* G;
    const float* g;
    float sd, sd2, lognorm, dt;
    int du, dv, D;
};

template <int DMAX>
__device__ __forceinline__ StepTables<DMAX> step_tables(const LgDev& d, int s) {
    StepTables<DMAX> t;
    t.G = d.G + (size_t)s * d.D * d.D;
    t.g = d.g + (size_t)s * d.D;
    t.sd = d.sd[s];
    t.sd2 = t.sd * t.sd;
    t.lognorm = d.lognorm[s];
    t.dt = d.dt;
    t.du = d.du;
    t.dv = d.dv;
    t.D = d.D;
    return t;
}

// drift_r = g_r + sum_c G_rc z_c, a c-ordered fma chain started at g_r, z = (u, v_prev)
template <int DMAX>
__device__ __forceinline__ float drift_row(const StepTables<DMAX>& t, int r, const float (&u)[DMAX],
                                           const float* __restrict__ v_prev) {
    const float* Gr = t.G + (size_t)r * t.D;
    float acc = t.g[r];
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
        if (c < t.du) acc = fbsmi_fmaf(Gr[c], u[c], acc);
#pragma unroll
    for (int c = 0; c < DMAX; ++c)
        if (c < t.dv) acc = fbsmi_fmaf(Gr[t.du + c], v_prev[c], acc);
    return acc;
}

__device__ __forceinline__ float norm_logpdf(float x, float loc, float sd2, float lognorm) {
    const float dlt = x - loc;
    return (lognorm + (dlt * dlt) / sd2) / -2.0f;
}

// likelihood_logpdf(v, u_prev, v_prev, t_prev): gp_gibbs.py:131-135
template <int DMAX>
__device__ __forceinline__ float lg_loglik(const StepTables<DMAX>& t, const float (&u)[DMAX],
                                           const float* __restrict__ v, const float* __restrict__ v_prev) {
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
        if (r < t.dv) {
            const float dr = drift_row<DMAX>(t, t.du + r, u, v_prev);
            const float cond_m = v_prev[r] + dr * t.dt;
            const float lp = norm_logpdf(v[r], cond_m, t.sd2, t.lognorm);
            acc = r == 0 ? lp : acc + lp;
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------
// init: csmc.py:150-155 with the init_sampler / init_likelihood_logpdf of gibbs.py:132-144
// ------------------------------------------------------------------------------------------
template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_init(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    const StepTables<DMAX> t = step_tables<DMAX>(d, 0);
    const int b0 = d.bs[0];
    const int base = (blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float lv[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = base + i;
        lv[i] = -__builtin_inff();
        if (p < d.N) {
            float u[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                if (r < d.du) {
                    float v = d.us_star[r];
                    if (d.ef && p != b0)
                        v = normal_at(d.misc[2], d.misc[3], (uint64_t)d.N * d.du, (uint64_t)p * d.du + r);
                    u[r] = v;
                    d.u0[(size_t)r * d.N + p] = v;
                    if (d.uss) d.uss[(size_t)p * d.du + r] = v;
                } else {
                    u[r] = 0.0f;
                }
            }
            // gibbs.py:136-137: likelihood_logpdf(vs[0], u0s, vs[1], ts[0]) ; :143-144: -log(nparticles)
            const float l = d.ef ? lg_loglik<DMAX>(t, u, d.vs, d.vs + d.dv) : d.lw_init;
            d.lw[p] = l;
            lv[i] = l;
        }
    }
    float m, sx;
    block_lse_partial<ITEMS>(lv, xch[0], xch[1], m, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = m;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// ------------------------------------------------------------------------------------------
// The three step kernels.  Each is latency-bound (a few hundred bytes per workgroup), so they are
// written to keep the dependent chain short: every global load whose address is known is issued
// at entry, reductions that are independent share one LDS exchange (block_upsweep_n), the
// per-workgroup partials of the previous kernel are re-reduced through a tile-shaped top tree
// (top_load / top_leaf) and searches use bisect_heap.
// ------------------------------------------------------------------------------------------
// A thread's chunk of ITEMS consecutive elements (the leaves of its subtree in the canonical summation tree) as 16-byte
// accesses when the whole chunk exists: one guarded dword per element -- what the straightforward loop compiles to, the
// guard forbids merging -- makes every wave instruction touch 64 separate cache lines, and the several-slots-per-thread
// kernels (N > 131072) then run at a tenth of the memory rate (measured: k_lg_norm<16> 0.7 TB/s).  Per-chain arrays are
// only 4-byte aligned (N is arbitrary), hence the vector type's alignment.
typedef float lg_f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int ITEMS>
__device__ __forceinline__ void chunk_load(const float* __restrict__ p, int base, int n, float fill, float (&x)[ITEMS]) {
    if (ITEMS % 4 == 0 && base + ITEMS <= n) {
#pragma unroll
        for (int i = 0; i < ITEMS; i += 4) {
            const lg_f4u q = *reinterpret_cast<const lg_f4u*>(p + base + i);
            x[i] = q.x; x[i + 1] = q.y; x[i + 2] = q.z; x[i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) x[i] = base + i < n ? p[base + i] : fill;
    }
}

template <int ITEMS>
__device__ __forceinline__ void chunk_store(float* __restrict__ p, int base, int n, const float (&x)[ITEMS]) {
    if (ITEMS % 4 == 0 && base + ITEMS <= n) {
#pragma unroll
        for (int i = 0; i < ITEMS; i += 4) {
            lg_f4u q;
            q.x = x[i]; q.y = x[i + 1]; q.z = x[i + 2]; q.w = x[i + 3];
            *reinterpret_cast<lg_f4u*>(p + base + i) = q;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (base + i < n) p[base + i] = x[i];
    }
}

__device__ __forceinline__ float block_max4(float m, float* lds4) {  // lds4 untouched since the last barrier
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}

// ------------------------------------------------------------------------------------------
// norm.  MODE 0: step s (J_prob partials for the conditional killing of step s);
//        MODE 1: final, explicit backward (force_move rest-weight partials, gibbs.py:200-205);
//        MODE 2: final, backward scanning (partials of w only).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float jprob_at(float w, float w_max, int N) { return (1.0f - w / w_max) / (float)N; }
// the same value when N is a power of two and inv_n = 1 / N (scaling by a power of two is exact, and rounds the same way
// into the subnormals)
__device__ __forceinline__ float jprob_pow2(float w, float w_max, float inv_n) { return (1.0f - w / w_max) * inv_n; }

__device__ __forceinline__ float fm_rest_at(float w, float w_k, bool is_k, int N) {
    if (w_k < 1.0f) return (is_k ? 0.0f : w) / (1.0f - w_k);
    return (float)(1.0 / (double)N);
}

// The node of the 256-leaf heap (root = 1) whose right half starts at leaf i >= 1, and the sum of that node's left
// half out of leaf i's up-sweep record: leaf i with c trailing zero bits is the midpoint of the node of 2^(c+1) leaves
// that contains it.
__device__ __forceinline__ int tree_mid_node(int i) {
    const int c = __builtin_ctz(i);
    return (1 << (7 - c)) + (i >> (c + 1));
}

__device__ __forceinline__ float tree_left_sum(const TreePath p, int i) {   // by value: see below
    // the lowest set bit of i picks the level; written as bit tests so that the record stays in registers (an index
    // computed from ctz(i) becomes a load from scratch memory)
    float v = p.ls[7];
    v = (i & 64) ? p.ls[6] : v;
    v = (i & 32) ? p.ls[5] : v;
    v = (i & 16) ? p.ls[4] : v;
    v = (i & 8) ? p.ls[3] : v;
    v = (i & 4) ? p.ls[2] : v;
    v = (i & 2) ? p.ls[1] : v;
    v = (i & 1) ? p.ls[0] : v;
    return v;
}

// the sum of the block of 2^lv leaves next to this thread's own block of 2^lv leaves (its sibling in the tile's tree):
// levels 0..5 inside the wave, 6 the other wave of the pair, 7 the other pair of waves
__device__ __forceinline__ float tree_sibling_sum(const TreePath& p, int lv) {
    if (lv < 7) return p.ls[lv];
    return (threadIdx.x & 128) ? p.ls[7] : p.hi;
}

// the tile's tree sum when the leaf of the thread that recorded `sib` (its eight sibling sums, leaf level first) holds x
__device__ __forceinline__ float tree_fold(float x, const float (&sib)[8], int levels = 8) {
#pragma unroll
    for (int lv = 0; lv < 8; ++lv) x = lv < levels ? x + sib[lv] : x;
    return x;
}

// One-tile wide Gibbs step: the draws of step sn that do not depend on the step before it -- normal(key_transition, (N, du)) and the
// two uniforms per source slot of the conditional killing resampler -- by producer block `blk` of `nblk` (idle blocks of a pinned
// launch, extra blocks of an unpinned one, or every working block a share at its end), into slot sn & 1.
__device__ __forceinline__ void wide_draw_ahead(const LgDev& d, int sn, int blk, int nblk) {
    const uint32_t* kt = d.keytab + 8 * sn;
    const uint32_t n0 = kt[6], n1 = kt[7];
    const int total = d.N * d.du;
    float* dst = d.xiw + (size_t)(sn & 1) * total;
    for (int e = blk * kBlock + (int)threadIdx.x; e < total; e += nblk * kBlock) dst[e] = normal_at(n0, n1, (uint64_t)total, (uint64_t)e);
    if (d.uw) {   // (from the far end of the producers: the first ones hold the larger noise shares)
        const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3];
        float* du2 = d.uw + (size_t)(sn & 1) * 2 * d.N;
        for (int e = (nblk - 1 - blk) * kBlock + (int)threadIdx.x; e < 2 * d.N; e += nblk * kBlock)
            du2[e] = e < d.N ? uniform_at(a0, a1, (uint64_t)d.N, (uint64_t)e) : uniform_at(b0, b1, (uint64_t)d.N, (uint64_t)(e - d.N));
    }
}

// One third (`part` of 3) of normal(key_transition, (N, du)) of step s into d.xiw, by the `nblk` extra blocks of a launch
// whose own work needs a few dozen workgroups: elements a and a + n/2 of the draw are the two words of one Threefry call
// (jax's random_bits), so a thread that owns a pair draws two normals per block-cipher call.
__device__ __forceinline__ void wide_noise_share(const LgDev& d, int s, uint32_t part, uint32_t blk, uint32_t nblk) {
    const uint32_t t0 = d.keytab[8 * s + 6], t1 = d.keytab[8 * s + 7];
    const uint32_t n = (uint32_t)d.N * (uint32_t)d.du, half = (n + 1u) >> 1, third = (half + 2u) / 3u;
    const uint32_t lo = part * third, hi = lo + third < half ? lo + third : half;
    for (uint32_t a = lo + blk * kBlock + threadIdx.x; a < hi; a += nblk * kBlock) {
        const uint32_t b = a + half;
        uint32_t o0, o1;
        threefry2x32(t0, t1, a, b < n ? b : 0u, o0, o1);
        d.xiw[a] = normal_from_bits(o0);
        if (b < n) d.xiw[b] = normal_from_bits(o1);
    }
}

template <int ITEMS, int MODE, bool PUB = false>
__global__ void __launch_bounds__(kBlock) k_lg_norm(LgDev dd, int s) {
    if (ITEMS == 1 && MODE == 0 && !PUB && dd.nz && (int)blockIdx.x >= dd.nb) {
        wide_noise_share(chain_view(dd, blockIdx.y), s, 0, blockIdx.x - dd.nb, gridDim.x - dd.nb);
        return;
    }
    int bx = blockIdx.x;
    if (PUB && dd.pin) {
        if (blockIdx.x & 7) return;
        bx = blockIdx.x >> 3;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    if (MODE == 0) { FBSMI_STAMP(2) }
    if (MODE == 0 && PUB) { FBSMI_SPAN_IN(0, s, 4) }
    __shared__ float xch[4][4];
    const int base = (bx * kBlock + threadIdx.x) * ITEMS;
    const int i_ref = d.bs[MODE == 0 ? s : d.T];
    float l[ITEMS];
    chunk_load<ITEMS>(d.lw, base, d.N, 0.0f, l);
    const float l_ref = MODE == 1 ? d.lw[i_ref] : 0.0f;
    // two-level logsumexp: combine the per-workgroup (max, sumexp) pairs the previous kernel published
    float lse, Mraw;
    lse_from_partials(d.bmax, d.bsumexp, d.nb, xch[0], xch[1], lse, Mraw);
    if (MODE == 0) { FBSMI_STAMP(14) }
    const float w_max = fbsmi_expf(Mraw - lse);  // == max_i w_i: fbsmi_expf is monotone
    const float w_k = MODE == 1 ? fbsmi_expf(l_ref - lse) : 0.0f;
    float xw[ITEMS], xj[ITEMS];
    const float inv_n = 1.0f / (float)d.N;   // PUB: N is a power of two, x / N == x * (1 / N) exactly
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int e = base + i;
        xw[i] = 0.0f;
        xj[i] = 0.0f;
        const bool in = e < d.N;
        l[i] = in ? l[i] - lse : 0.0f;       // from here on: the normalised log-weight
        if (in) {
            const float w = fbsmi_expf(l[i]);
            xw[i] = w;
            if (MODE == 0) xj[i] = e == i_ref ? 0.0f : ((PUB && !d.plus1) ? jprob_pow2(w, w_max, inv_n) : jprob_at(w, w_max, d.N));
            if (MODE == 1) xj[i] = fm_rest_at(w, w_k, e == i_ref, d.N);
        }
        // several elements per thread: left alone, the scheduler interleaves all sixteen float64 exponentials and their
        // divisions and the kernel needs 510 registers (one wave per SIMD; measured 273 us for 8M elements).  Four at a time.
        if (ITEMS > 4 && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    chunk_store<ITEMS>(d.w, base, d.N, xw);
    if (MODE != 0) chunk_store<ITEMS>(d.lwn, base, d.N, l);   // read back only after the final normalisation (view 1)
    if (d.lwss) chunk_store<ITEMS>(d.lwss + (size_t)s * d.N, base, d.N, l);
    float s2[2] = {chunk_total<ITEMS>(xw), chunk_total<ITEMS>(xj)}, t2[2];
    TreePath p2[2];
    block_upsweep_n<2>(s2, p2, xch[2], t2);
    if (PUB && ITEMS == 1) {   // the tile's part of the summation tree, for the tree-walking searches of k_lg_prop1t
        if (MODE == 0 && base == i_ref) {
            // The J_prob leaf of the reference index is 0 here and becomes J_prob[i*] = 1 - (sum of the others) once every
            // tile's sum is known (resamplings.py:80-82).  A changed leaf changes the tree sums on its path only, and each of
            // them is (sum below) + (sibling's sum): this thread's eight sibling sums are all the next kernel needs to redo
            // the tile's sum for ANY value of that leaf -- it no longer fetches the tile and re-reduces it.
#pragma unroll
            for (int lv = 0; lv < 8; ++lv) d.scal[4 + lv] = tree_sibling_sum(p2[1], lv);
        }
        const int i = threadIdx.x;
        if ((i & 3) == 0) {   // the midpoints of the nodes of 8 leaves and more: 64 threads, 16 per wave
            const int h = i ? tree_mid_node(i) : 0;
            const float2 node = make_float2(i ? tree_left_sum(p2[0], i) : 0.0f, xw[0]);
            d.trW[(size_t)bx * kTreeNodes + h] = node;
            if (h < kMidN) d.trWtop[bx * kMidN + h] = node;
            if (i == 0) d.wfirst[bx] = node.y;
        }
    }
    if (threadIdx.x == 0) {
        if (MODE != 1) d.bsumw[bx] = t2[0];
        if (MODE != 2) d.bsumJ[bx] = t2[1];
        if (bx == 0) {
            d.scal[0] = lse;
            d.scal[1] = w_max;
            d.scal[2] = w_k;
        }
    }
    if (MODE == 0) { FBSMI_STAMP(3) }
    if (MODE == 0 && PUB) { FBSMI_SPAN_OUT(0, s, 4) }
}

// (Round 3 also built this kernel with ONE WAVE PER TILE -- four tiles per workgroup, four consecutive elements per lane, the
// chunk sum as levels 0-1 of the tile tree and the wave's butterfly as levels 2-7, no exchange between waves, the logsumexp combine
// once per four tiles; bit-exact -- on the theory that launching a quarter of the waves would shorten the launch.  It does not:
// the workgroups of a launch enter over 0.6-1.1 us whether there are 64 or 256 of them (per-workgroup stamps, tools/stamps), and a
// lone wave issues one instruction per ~4.4 clocks, so ~1000 instructions in one wave instead of ~370 in each of four cost
// 3.7 us after the normaliser is known instead of 0.9.  One chain 22.7 against 13.0 us per step, 32 chains 83 against 52.
// Dropped: on this chip latency wants MORE, thinner waves, not fewer.)
// ------------------------------------------------------------------------------------------
// cdf.  MODE 0: cumsum(w) -> cdf and cumsum(J_prob) -> cdfJ; MODE 1: cumsum(rest) -> cdf;
//       MODE 2: cumsum(w) -> cdf.
// ------------------------------------------------------------------------------------------
template <int ITEMS, int MODE>
__global__ void __launch_bounds__(kBlock) k_lg_cdf(LgDev dd, int s) {
    if (ITEMS == 1 && MODE == 0 && dd.nz && (int)blockIdx.x >= dd.nb) {
        wide_noise_share(chain_view(dd, blockIdx.y), s, 1, blockIdx.x - dd.nb, gridDim.x - dd.nb);
        return;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    if (MODE == 0) { FBSMI_STAMP(4) }
    __shared__ float xch[8][4];
    __shared__ float bc[4][2];
    constexpr int TILE = kBlock * ITEMS;
    const int b = blockIdx.x;
    const int i_ref = d.bs[MODE == 0 ? s : d.T];
    const int base = (b * kBlock + threadIdx.x) * ITEMS;
    // ---- everything addressable now is loaded now
    const float w_max = d.scal[1];
    const float w_k = d.scal[2];
    float wv[ITEMS];
    chunk_load<ITEMS>(d.w, base, d.N, 0.0f, wv);
    if (MODE == 1) {
        float pj[kTopItems];
        top_load(d.bsumJ, d.nb, pj);
        float x[ITEMS], c[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) x[i] = base + i < d.N ? fm_rest_at(wv[i], w_k, base + i == i_ref, d.N) : 0.0f;
        float s2[2] = {chunk_total<kTopItems>(pj), chunk_total<ITEMS>(x)}, t2[2];
        TreePath p2[2];
        block_upsweep_n<2>(s2, p2, xch[0], t2);
        float P, E;
        top_leaf(pj, p2[0], t2[0], b, bc[0], P, E);
        block_descend(P, E, p2[1]);
        chunk_scan<ITEMS>(x, P, E, c);
        chunk_store<ITEMS>(d.cdf, base, d.N, c);
        return;
    }
    float pw[kTopItems];
    top_load(d.bsumw, d.nb, pw);
    if (MODE == 2) {
        float c[ITEMS];
        float s2[2] = {chunk_total<kTopItems>(pw), chunk_total<ITEMS>(wv)}, t2[2];
        TreePath p2[2];
        block_upsweep_n<2>(s2, p2, xch[0], t2);
        float P, E;
        top_leaf(pw, p2[0], t2[0], b, bc[0], P, E);
        block_descend(P, E, p2[1]);
        chunk_scan<ITEMS>(wv, P, E, c);
        chunk_store<ITEMS>(d.cdf, base, d.N, c);
        return;
    }
    // ---- MODE 0
    // Which node of the implicit bisection tree is this element the midpoint of (if any, within the heap
    // depth)?  A property of (N, element) alone: tabulated once at handle creation (heap_map()).
    int hp_node = 0, hp_depth = 0;
    if (ITEMS == 1 && d.hpW && base < d.N) {
        const int hm = d.hp_map[base];
        hp_node = hm & 0xFFFF;
        hp_depth = hm >> 16;
    }
    const int b_ref = i_ref / TILE;
    const int rbase = b_ref * TILE + threadIdx.x * ITEMS;
    float wr[ITEMS];
    chunk_load<ITEMS>(d.w, rbase, d.N, 0.0f, wr);
    float pj[kTopItems];
    top_load(d.bsumJ, d.nb, pj);
    // phase 1: top(bsumw), top(bsumJ) [for the total], own w tile -- one exchange
    float s3[3] = {chunk_total<kTopItems>(pw), chunk_total<kTopItems>(pj), chunk_total<ITEMS>(wv)}, t3[3];
    TreePath p3[3];
    block_upsweep_n<3>(s3, p3, xch[0], t3);
    FBSMI_STAMP(15)
    // J_prob[i*] = max(1 - sum(J_prob with [i*] = 0), 0)   (resamplings.py:80-82)
    const float Ji = fmaxf(1.0f - t3[1], 0.0f);
    // phase 2: the tile that holds i* (its tree sum changes) and this workgroup's own J tile
    float xr[ITEMS], xo[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int er = rbase + i, eo = base + i;
        xr[i] = er < d.N ? (er == i_ref ? Ji : jprob_at(wr[i], w_max, d.N)) : 0.0f;
        xo[i] = eo < d.N ? (eo == i_ref ? Ji : jprob_at(wv[i], w_max, d.N)) : 0.0f;
    }
    float s2[2] = {chunk_total<ITEMS>(xr), chunk_total<ITEMS>(xo)}, t2[2];
    TreePath p2[2];
    block_upsweep_n<2>(s2, p2, xch[3], t2);
    // phase 3: top tree of the J partials with the rebuilt tile substituted
    float pj2[kTopItems];
#pragma unroll
    for (int i = 0; i < kTopItems; ++i) pj2[i] = (int)threadIdx.x * kTopItems + i == b_ref ? t2[0] : pj[i];
    float s1[1] = {chunk_total<kTopItems>(pj2)}, t1[1];
    TreePath p1[1];
    block_upsweep_n<1>(s1, p1, xch[5], t1);
    // (P, E) of this workgroup's tile in both trees (one barrier for both broadcasts)
    {
        float pw_p = 0.0f, pw_e = t3[0], pj_p = 0.0f, pj_e = t1[0];
        block_descend(pw_p, pw_e, p3[0]);
        block_descend(pj_p, pj_e, p1[0]);
        if ((int)threadIdx.x == (b >> 2)) {
            float t = pw_p + (pw[0] + pw[1]);
            if (b & 2) pw_p = t; else pw_e = t;
            t = pw_p + ((b & 2) ? pw[2] : pw[0]);
            if (b & 1) pw_p = t; else pw_e = t;
            bc[0][0] = pw_p;
            bc[0][1] = pw_e;
            t = pj_p + (pj2[0] + pj2[1]);
            if (b & 2) pj_p = t; else pj_e = t;
            t = pj_p + ((b & 2) ? pj2[2] : pj2[0]);
            if (b & 1) pj_p = t; else pj_e = t;
            bc[1][0] = pj_p;
            bc[1][1] = pj_e;
        }
        __syncthreads();
    }
    FBSMI_STAMP(16)
    float P = bc[0][0], E = bc[0][1];
    float c[ITEMS];
    block_descend(P, E, p3[2]);
    chunk_scan<ITEMS>(wv, P, E, c);
    chunk_store<ITEMS>(d.cdf, base, d.N, c);
    if (ITEMS == 1 && hp_node) d.hpW[hp_node] = c[0];
    P = bc[1][0];
    E = bc[1][1];
    block_descend(P, E, p2[1]);
    chunk_scan<ITEMS>(xo, P, E, c);
    chunk_store<ITEMS>(d.cdfJ, base, d.N, c);
    if (ITEMS == 1 && hp_node && hp_depth < d.lh_j) d.hpJ[hp_node] = c[0];
    FBSMI_STAMP(5)
}

// ------------------------------------------------------------------------------------------
// prop: resample (killing, conditional) + gather + Euler-Maruyama + pin + log-weight
// ------------------------------------------------------------------------------------------
template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_prop(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    __shared__ float heapW[kHeapSize], heapJ[kHeapSize];
    const int N = d.N;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    float hw = 0.0f, hj = 0.0f;
    if (threadIdx.x >= 1 && threadIdx.x < kHeapSize) {
        const int mid = heap_node_mid(threadIdx.x, N);
        hw = d.cdf[mid];
        hj = d.cdfJ[mid];
    }
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const int base = (blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    // the slot's own noise depends on nothing that is still in flight: draw it while the heap
    // gathers are outstanding
    constexpr bool kHoistNoise = ITEMS * DMAX <= 16;   // more than that would spill
    float xi[kHoistNoise ? ITEMS : 1][kHoistNoise ? DMAX : 1];
    const float u3 = __uint_as_float(kt[4]);
    if (kHoistNoise) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
#pragma unroll
            for (int r = 0; r < DMAX; ++r)
                xi[kHoistNoise ? i : 0][kHoistNoise ? r : 0] =
                    (r < d.du && base + i < N) ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)(base + i) * d.du + r)
                                              : 0.0f;
    }
    if (threadIdx.x < kHeapSize) {
        heapW[threadIdx.x] = hw;
        heapJ[threadIdx.x] = hj;
    }
    __syncthreads();
    // J = choice(key_3, N, (), p=J_prob)  (resamplings.py:84); roll by j - J (:85).  Every thread
    // repeats the (identical, broadcast-served) search: no further barrier is needed.
    int shift;
    {
        const int J = bisect_heap(d.cdfJ, N, d.levels, heapJ, lastJ * (1.0f - u3));
        shift = (j_ref - J) % N;
        if (shift < 0) shift += N;
    }
    float lv[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int m = base + i;
        lv[i] = -__builtin_inff();
        if (m < N) {
            int src = m - shift;
            if (src < 0) src += N;
            const float ws = d.w[src];
            const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src);
            int a = src;
            if (u1 * w_max >= ws) {  // killed (resamplings.py:71): redraw from Cat(w) (:73-74)
                const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)src);
                a = bisect_heap(d.cdf, N, d.levels, heapW, last * (1.0f - u2));
            }
            if (m == j_ref) a = i_ref;  // :86
            if (d.As) d.As[(size_t)s * N + m] = a;
            float u[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + a] : 0.0f;
            // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                if (r < d.du) {
                    const float dr = drift_row<DMAX>(t, r, u, v_prev);
                    const float z = kHoistNoise ? xi[kHoistNoise ? i : 0][kHoistNoise ? r : 0]
                                                : normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)m * d.du + r);
                    float x = (u[r] + dr * t.dt) + t.sd * z;
                    if (m == j_ref) x = ustar[r];
                    un[(size_t)r * N + m] = x;
                    if (d.uss) d.uss[((size_t)(s + 1) * N + m) * d.du + r] = x;
                }
            }
            // likelihood_logpdf on the gathered particle (csmc.py:145)
            const float l = lg_loglik<DMAX>(t, u, v, v_prev);
            d.lw[m] = l;
            lv[i] = l;
        }
    }
    float mx, sx;
    block_lse_partial<ITEMS>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// ------------------------------------------------------------------------------------------
// prop, one slot per thread (N <= 131072): the same step as k_lg_prop, arranged around what bounds
// it -- the number of dependent memory round trips when one chain runs alone, and the number of
// scattered (one cache line per lane) loads when several chains fill the CUs:
//   0. everything addressable at entry: the compact heaps the cdf kernel published (coalesced:
//      2^11 nodes of cdf, 2^8 of cdfJ), tables, the reference row; the slot's own noise is drawn in
//      the shadow of these loads;
//   1. J: LDS levels, then the workgroup fetches the remaining interval of cdfJ whole (coalesced);
//   2. w[src] and the source's own row (a survivor is its own ancestor), rotated but coalesced;
//   3-4. killed slots only: 11 levels of the Cat(w) search in LDS, the rest three levels per
//      round trip;  5. killed slots only: the ancestor row.
// ------------------------------------------------------------------------------------------
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_prop1(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    __shared__ __attribute__((aligned(16))) float heapW[kHeapSizeW];
    __shared__ float heapJ[kHeapSizeJ];
    __shared__ float win[kBlock];
    FBSMI_STAMP(6)
    const int N = d.N;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const bool live = m < N;
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    // the compact heaps, whole (nodes past the published depth are zero and never visited): two float4 per thread
    static_assert(kHeapSizeW == 8 * kBlock && kHeapSizeJ == kBlock, "heap staging assumes 2048 / 256 nodes");
    const float4 hw0 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x];
    const float4 hw1 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x + 1];
    const float hj = d.hpJ[threadIdx.x];
    float uref[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) uref[r] = r < d.du ? up[(size_t)r * N + i_ref] : 0.0f;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    float xi[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r)
        xi[r] = (r < d.du && live) ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)m * d.du + r) : 0.0f;
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x] = hw0;
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x + 1] = hw1;
    heapJ[threadIdx.x] = hj;
    FBSMI_STAMP(7)
    __syncthreads();
    FBSMI_STAMP(8)
    // ---- round 1: J = choice(key_3, N, (), p=J_prob) (resamplings.py:84); roll by j - J (:85)
    const int J = bisect_uniform(d.cdfJ, N, d.levels, d.lh_j, heapJ, win, lastJ * (1.0f - u3));
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
    int src = m - shift;
    if (src < 0) src += N;
    if (!live) src = 0;
    FBSMI_STAMP(9)
    // ---- round 2
    const float ws = d.w[src];
    float u[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + src] : 0.0f;
    const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src);
    const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)src);
    const float qK = last * (1.0f - u2);                                    // resamplings.py:73-74
    int lo, hi;
    bisect_lds_levels(N, d.lh_w, heapW, qK, lo, hi);
    const bool killed = live && (u1 * w_max >= ws);                         // :71
    FBSMI_STAMP(10)
    // ---- rounds 3, 4
#pragma unroll 1
    for (int rem = d.levels - d.lh_w; rem > 0; rem -= 3) bisect_round3(d.cdf, lo, hi, qK, killed);
    const bool pinned = m == j_ref;
    const int a = pinned ? i_ref : (killed ? hi : src);                     // :86
    FBSMI_STAMP(11)
    // ---- round 5
    if (killed && !pinned) {
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            if (r < d.du) u[r] = up[(size_t)r * N + a];
    }
    if (pinned) {
#pragma unroll
        for (int r = 0; r < DMAX; ++r) u[r] = uref[r];
    }
    float lv[1] = {-__builtin_inff()};
    if (live) {
        if (d.As) d.As[(size_t)s * N + m] = a;
        // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            if (r < d.du) {
                const float dr = drift_row<DMAX>(t, r, u, v_prev);
                float x = (u[r] + dr * t.dt) + t.sd * xi[r];
                if (pinned) x = ustar[r];
                un[(size_t)r * N + m] = x;
                if (d.uss) d.uss[((size_t)(s + 1) * N + m) * d.du + r] = x;
            }
        }
        const float l = lg_loglik<DMAX>(t, u, v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
        d.lw[m] = l;
        lv[0] = l;
    }
    FBSMI_STAMP(12)
    float mx, sx;
    block_lse_partial<1>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
    FBSMI_STAMP(13)
}

// ------------------------------------------------------------------------------------------
// Several slots per thread (N > 131072: tiles of 1024 / 4096 slots, ITEMS = 4 / 16), the HBM-resident regime.
// k_lg_prop above walks a thread's slots one after the other, each with its own chain of dependent probes (the search of
// a killed slot alone is five global round trips behind eight LDS levels) and with the thread's ITEMS slots CONSECUTIVE
// in memory -- a wave's load touches 64 cache lines.  Measured at 4 x 2^22 particles: 1.3 ms per step, 0.05 of the HBM
// roof, the CUs waiting on ~160 serial round trips per thread.  Here:
//   * slot i of thread t is element tile0 + i * 256 + t: every access of a wave is one contiguous run;
//   * the slots are taken four at a time, all loads of a stage issued before the first is used, the four searches in
//     lockstep (bisect_round3_xn): a batch costs the round trips of ONE slot;
//   * the compact heaps of both CDFs (11 / 8 levels) come from k_lg_heaps, a launch of eight workgroups behind k_lg_cdf,
//     so a workgroup stages them with two coalesced loads per thread and a search leaves LDS with 2048 candidates left;
//   * the tile's (max, sumexp) wants the canonical chunk order (thread t owns elements t * ITEMS ..): the new log-weights
//     cross LDS once.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_lg_heaps(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const int t = blockIdx.x * kBlock + threadIdx.x;   // node of the implicit bisection tree (root 1); grid.x = kHeapSizeW / kBlock
    if (t < 1 || t >= kHeapSizeW) return;
    const int depth = 31 - __builtin_clz(t);
    if (depth >= d.lh_w) return;
    const int mid = heap_node_mid(t, d.N);
    d.hpW[t] = d.cdf[mid];
    if (t < kHeapSizeJ && depth < d.lh_j) d.hpJ[t] = d.cdfJ[mid];
}

template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_propN(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    constexpr int B = ITEMS < 4 ? ITEMS : 4;       // slots in flight per thread
    constexpr int TILE = kBlock * ITEMS;
    __shared__ float xch[2][4];
    __shared__ __attribute__((aligned(16))) float heapW[kHeapSizeW];
    __shared__ float heapJ[kHeapSizeJ];
    __shared__ float win[kBlock];
    __shared__ __attribute__((aligned(16))) float lvs[TILE];
    const int N = d.N;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int tile0 = blockIdx.x * TILE;
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    static_assert(kHeapSizeW == 8 * kBlock && kHeapSizeJ == kBlock, "heap staging assumes 2048 / 256 nodes");
    const float4 hw0 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x];
    const float4 hw1 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x + 1];
    const float hj = d.hpJ[threadIdx.x];
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x] = hw0;
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x + 1] = hw1;
    heapJ[threadIdx.x] = hj;
    __syncthreads();
    // ---- round 1: J = choice(key_3, N, (), p=J_prob) (resamplings.py:84); roll by j - J (:85)
    const int J = bisect_uniform(d.cdfJ, N, d.levels, d.lh_j, heapJ, win, lastJ * (1.0f - u3));
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
#pragma unroll 1
    for (int i0 = 0; i0 < ITEMS; i0 += B) {
        int m[B], src[B];
        bool live[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            m[k] = tile0 + (i0 + k) * kBlock + (int)threadIdx.x;
            live[k] = m[k] < N;
            int sc = m[k] - shift;
            if (sc < 0) sc += N;
            src[k] = live[k] ? sc : 0;
        }
        // ---- round 2: the sources' weights, rotated but contiguous
        float ws[B], u[B][DMAX];
#pragma unroll
        for (int k = 0; k < B; ++k) ws[k] = d.w[src[k]];
        float qK[B], xi[B][DMAX];
        bool killed[B];
        int lo[B], hi[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src[k]);
            const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)src[k]);
#pragma unroll
            for (int r = 0; r < DMAX; ++r)
                xi[k][r] = r < d.du ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)(live[k] ? m[k] : 0) * d.du + r) : 0.0f;
            qK[k] = last * (1.0f - u2);                                      // resamplings.py:73-74
            bisect_lds_levels(N, d.lh_w, heapW, qK[k], lo[k], hi[k]);
            killed[k] = live[k] && (u1 * w_max >= ws[k]);                     // :71
        }
        // ---- rounds 3..: the killed slots' searches, in lockstep
#pragma unroll 1
        for (int rem = d.levels - d.lh_w; rem > 0; rem -= 3) bisect_round3_xn<B>(d.cdf, lo, hi, qK, killed);
        // ---- the ancestors' rows (a survivor's is its own source: contiguous; no load under a divergent branch)
        int a[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const bool pinned = m[k] == j_ref;
            a[k] = pinned ? i_ref : (killed[k] ? hi[k] : src[k]);            // :86
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[k][r] = r < d.du ? up[(size_t)r * N + a[k]] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const bool pinned = m[k] == j_ref;
            float l = -__builtin_inff();
            if (live[k]) {
                if (d.As) d.As[(size_t)s * N + m[k]] = a[k];
                // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
                for (int r = 0; r < DMAX; ++r) {
                    if (r < d.du) {
                        const float dr = drift_row<DMAX>(t, r, u[k], v_prev);
                        float x = (u[k][r] + dr * t.dt) + t.sd * xi[k][r];
                        if (pinned) x = ustar[r];
                        un[(size_t)r * N + m[k]] = x;
                        if (d.uss) d.uss[((size_t)(s + 1) * N + m[k]) * d.du + r] = x;
                    }
                }
                l = lg_loglik<DMAX>(t, u[k], v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
                d.lw[m[k]] = l;
            }
            lvs[(i0 + k) * kBlock + threadIdx.x] = l;
        }
    }
    __syncthreads();
    float lv[ITEMS];
    if (ITEMS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; i += 4) {
            const float4 q4 = *reinterpret_cast<const float4*>(lvs + threadIdx.x * ITEMS + i);
            lv[i] = q4.x; lv[i + 1] = q4.y; lv[i + 2] = q4.z; lv[i + 3] = q4.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) lv[i] = lvs[threadIdx.x * ITEMS + i];
    }
    float mx, sx;
    block_lse_partial<ITEMS>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// The same with the killed sources' searches COMPACTED: only ~1 - mean(w) / max(w) of the slots are killed (a few per cent
// when the weights are as flat as a short Euler step leaves them), yet in k_lg_propN every wave walks the 11 LDS levels and
// the four probe rounds for every one of its slots, because some lane of it usually has a killed one -- a third of that
// kernel's instructions.  Here the kill tests run first (phase 1) and push the killed slots of the tile into an LDS queue
// (one LDS atomic per wave and batch); phase 2 hands the queue out one entry per thread -- redraw uniform, LDS levels,
// probe rounds, all lanes busy --; phase 3 gathers, propagates and weights.  The ancestors found in phase 2 wait in the
// LDS array that afterwards carries the new log-weights to the chunk-ordered reduction (same owner thread per element).
template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_propQ(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    constexpr int B = ITEMS < 4 ? ITEMS : 4;       // slots in flight per thread
    constexpr int TILE = kBlock * ITEMS;
    static_assert(TILE <= 65536 && ITEMS <= 32, "queue entries are 16-bit local slots, kill flags one 32-bit word");
    __shared__ float xch[2][4];
    __shared__ __attribute__((aligned(16))) float heapW[kHeapSizeW];
    __shared__ float heapJ[kHeapSizeJ];
    __shared__ float win[kBlock];
    __shared__ __attribute__((aligned(16))) float lvs[TILE];
    __shared__ uint16_t queue[TILE];
    __shared__ int qn;
    const int N = d.N;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int tile0 = blockIdx.x * TILE;
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    static_assert(kHeapSizeW == 8 * kBlock && kHeapSizeJ == kBlock, "heap staging assumes 2048 / 256 nodes");
    const float4 hw0 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x];
    const float4 hw1 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x + 1];
    const float hj = d.hpJ[threadIdx.x];
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x] = hw0;
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x + 1] = hw1;
    heapJ[threadIdx.x] = hj;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    // ---- round 1: J = choice(key_3, N, (), p=J_prob) (resamplings.py:84); roll by j - J (:85)
    const int J = bisect_uniform(d.cdfJ, N, d.levels, d.lh_j, heapJ, win, lastJ * (1.0f - u3));
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
    // ---- phase 1: the kill tests (resamplings.py:71), the sources' weights rotated but contiguous
    uint32_t kmask = 0;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int i0 = 0; i0 < ITEMS; i0 += B) {
        int src[B];
        bool live[B];
        float ws[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int m = tile0 + (i0 + k) * kBlock + (int)threadIdx.x;
            live[k] = m < N;
            int sc = m - shift;
            if (sc < 0) sc += N;
            src[k] = live[k] ? sc : 0;
            ws[k] = d.w[src[k]];
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src[k]);
            const bool killed = live[k] && (u1 * w_max >= ws[k]);
            kmask |= (killed ? 1u : 0u) << (i0 + k);
            const unsigned long long bal = __ballot(killed);
            if (bal) {   // wave-uniform
                const int first = __ffsll((long long)bal) - 1;
                int qb = 0;
                if (lane == first) qb = atomicAdd(&qn, __popcll(bal));
                qb = __shfl(qb, first);
                if (killed) queue[qb + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)((i0 + k) * kBlock + (int)threadIdx.x);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: one queue entry per thread: the redraw from Cat(w) (:73-74)
    const int nq = qn;
    int* anc = reinterpret_cast<int*>(lvs);
    // two entries per thread and pass, in lockstep: a tile's queue is usually a little longer than the workgroup (~7 % of 4096
    // slots), and a second pass for its tail would cost the full chain of round trips again
#pragma unroll 1
    for (int e0 = 0; e0 < nq; e0 += 2 * kBlock) {
        bool on[2];
        int slot[2], lo[2], hi[2];
        float qK[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = e0 + k * kBlock + (int)threadIdx.x;
            on[k] = e < nq;
            slot[k] = on[k] ? (int)queue[e] : 0;
            int sc = tile0 + slot[k] - shift;
            if (sc < 0) sc += N;
            const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)(on[k] ? sc : 0));
            qK[k] = last * (1.0f - u2);
        }
        bisect_lds_levels_x2(N, d.lh_w, heapW, qK, lo, hi);
#pragma unroll 1
        for (int rem = d.levels - d.lh_w; rem > 0; rem -= 3) bisect_round3_xn<2>(d.cdf, lo, hi, qK, on);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (on[k]) anc[slot[k]] = hi[k];
    }
    __syncthreads();
    // ---- phase 3: ancestors' rows, Euler-Maruyama, pin, log-weight
#pragma unroll 1
    for (int i0 = 0; i0 < ITEMS; i0 += B) {
        int m[B], a[B];
        bool live[B];
        float u[B][DMAX], xi[B][DMAX];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int slot = (i0 + k) * kBlock + (int)threadIdx.x;
            m[k] = tile0 + slot;
            live[k] = m[k] < N;
            int sc = m[k] - shift;
            if (sc < 0) sc += N;
            const bool killed = (kmask >> (i0 + k)) & 1u;
            const int red = anc[slot];                 // (garbage unless killed)
            a[k] = !live[k] ? 0 : (m[k] == j_ref ? i_ref : (killed ? red : sc));          // :86
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[k][r] = r < d.du ? up[(size_t)r * N + a[k]] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r)
                xi[k][r] = r < d.du ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)(live[k] ? m[k] : 0) * d.du + r) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const bool pinned = m[k] == j_ref;
            float l = -__builtin_inff();
            if (live[k]) {
                if (d.As) d.As[(size_t)s * N + m[k]] = a[k];
                // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
                for (int r = 0; r < DMAX; ++r) {
                    if (r < d.du) {
                        const float dr = drift_row<DMAX>(t, r, u[k], v_prev);
                        float x = (u[k][r] + dr * t.dt) + t.sd * xi[k][r];
                        if (pinned) x = ustar[r];
                        un[(size_t)r * N + m[k]] = x;
                        if (d.uss) d.uss[((size_t)(s + 1) * N + m[k]) * d.du + r] = x;
                    }
                }
                l = lg_loglik<DMAX>(t, u[k], v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
                d.lw[m[k]] = l;
            }
            lvs[(i0 + k) * kBlock + threadIdx.x] = l;
        }
    }
    __syncthreads();
    float lv[ITEMS];
    if (ITEMS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; i += 4) {
            const float4 q4 = *reinterpret_cast<const float4*>(lvs + threadIdx.x * ITEMS + i);
            lv[i] = q4.x; lv[i + 1] = q4.y; lv[i + 2] = q4.z; lv[i + 3] = q4.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) lv[i] = lvs[threadIdx.x * ITEMS + i];
    }
    float mx, sx;
    block_lse_partial<ITEMS>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// ------------------------------------------------------------------------------------------
// The step in TWO launches (norm<PUB> -> k_lg_prop1t), for N a power of two with 2..256 tiles.
//
// The fixed-length bisection of searchsorted over the canonical cumsum IS a descent of the summation tree when N
// is a power of two: at a node X = [lo, lo + sz) with prefix P in front of it, the probe is
//      cdf[mid] = (P + sum(left half of X)) + x[mid]        (sz >= 4: leaf mid is the first leaf of the right half,
//                                                            a left child whose own prefix is t = P + sum(left))
//      cdf[mid] = E(X)                                      (sz == 2, and the closing one-leaf level)
// and going left / right hands (P, E = t) / (P = t, E) to the child -- the two-value descent of fbsmi_device.h.  So
// the searches need the tree's left-half sums and the leaves at the midpoints, never the cumsum itself: k_lg_norm
// publishes them per tile (the records of its own up-sweep), this kernel builds the levels above the tiles from the
// per-tile sums (which every workgroup re-reduces anyway), and the k_lg_cdf launch disappears from the step.
//   J (workgroup-uniform): tile sums -> total -> J_prob[i*] -> the rebuilt tile of i* -> top tree with its sum
//   substituted -> descent to a tile -> that tile's w fetched whole, its leaves and tree rebuilt in LDS -> descent.
//   Cat(w) per slot: top levels + three levels of every tile in LDS, then (killed slots only) three levels from the
//   tile's published heap in one round trip and the last four leaves of w in another.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tree_walk(float2 nd, float q, float& P, float& E) {
    const float t = P + nd.x;
    const bool gl = q <= t + nd.y;
    E = gl ? t : E;
    P = gl ? P : t;
    return gl;
}

// Two levels of a heap-ordered tree in ONE LDS round trip: the node and both children (adjacent: one 16-byte read) are
// fetched together.  Returns the heap index two levels down.
__device__ __forceinline__ int tree_walk2(const float2* heap, int h, float q, float& P, float& E) {
    const float2 a = heap[h];
    const float4 c = *reinterpret_cast<const float4*>(heap + 2 * h);
    const bool g1 = tree_walk(a, q, P, E);
    const bool g2 = tree_walk(g1 ? make_float2(c.x, c.y) : make_float2(c.z, c.w), q, P, E);
    return 4 * h + (g1 ? 0 : 2) + (g2 ? 0 : 1);
}

// `levels` levels from node h
__device__ __forceinline__ int tree_walk_n(const float2* heap, int h, int levels, float q, float& P, float& E) {
    for (; levels >= 2; levels -= 2) h = tree_walk2(heap, h, q, P, E);
    if (levels) h = 2 * h + (tree_walk(heap[h], q, P, E) ? 0 : 1);
    return h;
}

// what a workgroup of the two-launch step loads at entry for the trees (issued before the noise draws)
struct TreeEntry {
    float sw, sj, wf;       // tile `tid`: sum of w, sum of J_prob ([i*] = 0), w of its first element
    float refsib[8];        // sibling sums of the leaf i* in its tile's J_prob tree (published by k_lg_norm)
    float xw, xj;           // N = 2^k + 1: w and J_prob ([i*] = 0) of the last slot (the extra tile's sums); else 0
    float4 stg[kMidN / 2];  // this thread's part of trWtop
};

__device__ __forceinline__ TreeEntry tree_entry_loads(const LgDev& d, int i_ref) {
    const int tid = threadIdx.x, nb = d.nb - d.plus1;   // tiles of the power-of-two part
    TreeEntry e;
    const bool tl = tid < nb;
    e.sw = tl ? d.bsumw[tid] : 0.0f;
    e.sj = tl ? d.bsumJ[tid] : 0.0f;
    e.wf = tl ? d.wfirst[tid] : 0.0f;
    e.xw = d.plus1 ? d.bsumw[nb] : 0.0f;
    e.xj = d.plus1 ? d.bsumJ[nb] : 0.0f;
#pragma unroll
    for (int lv = 0; lv < 8; ++lv) e.refsib[lv] = d.scal[4 + lv];
#pragma unroll
    for (int k = 0; k < kMidN / 2; ++k) {
        const int idx = tid + kBlock * k;
        e.stg[k] = idx < kMidN / 2 * nb ? reinterpret_cast<const float4*>(d.trWtop)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return e;
}

// LDS of the tree walks: heaps over the 256 tile leaves for w and J_prob, the heap of the tile J falls in, and the
// first three levels of every tile's w heap
struct TreeLds {
    float xch[5][4];
    float bc[8];   // sibling sums of the tile of i* in the tree over the tiles (broadcast by the thread that owns that tile)
    __attribute__((aligned(16))) float2 topW[kBlock], topJ[kBlock], tileJ[kBlock];
    __attribute__((aligned(16))) float2 midW[kMidN * kBlock];
};

constexpr int kTreeBuildBarriers = 4;   // __syncthreads() executed by tree_build (two up-sweep exchanges, two explicit)

// Builds the heaps and finds J = choice(key_3, N, (), p=J_prob) (resamplings.py:84), the same in every thread;
// `last` = cdf[N - 1], `rootW` = the canonical inclusive prefix at the end of the power-of-two part (== last unless N = 2^k + 1).
// Ends with every heap visible to the workgroup.
//
// N = 2^k + 1 (LgDev.plus1).  The fixed-length bisection over 2^k + 1 elements probes, level by level, exactly the midpoints
// of the 2^k-leaf tree's nodes: an interval [a, a + 2^j + 1) on the right spine has its midpoint at a + 2^(j-1), the midpoint
// of the tree node [a, a + 2^j), and every left turn enters an aligned power-of-two block with the same number of levels left
// as in the 2^k walk.  The two walks differ only at the end of the all-right path, where the last level compares with
// cdf[2^k] = the total, which every query is below -- and both return 2^k there.  So the searches run over the tree of the
// first 2^k slots with queries scaled by the whole total; the last slot is an extra tile of one element that joins the sums.
__device__ __forceinline__ int tree_build(const LgDev& d, const TreeEntry& e, TreeLds& L, int i_ref, float u3, float w_max,
                                          float inv_n, float& last, float& rootW) {
    const int nb = d.nb - d.plus1, tid = threadIdx.x;
    const bool ref_extra = d.plus1 && i_ref == d.N - 1;   // the reference index is the last slot: nothing changes in the tree
    const int b_ref = ref_extra ? 0 : i_ref / kBlock;
#pragma unroll
    for (int k = 0; k < kMidN / 2; ++k) {
        const int idx = tid + kBlock * k;
        if (idx < kMidN / 2 * nb) reinterpret_cast<float4*>(L.midW)[idx] = e.stg[k];
    }
    // ---- the levels above the tiles: sums of w, sums of J_prob with [i*] = 0 (one exchange)
    const int g = tid ? tree_mid_node(tid) : 0;
    float s2[2] = {e.sw, e.sj}, t2[2];
    TreePath p2[2];
    // block_upsweep_n<2>, with one more thing riding on its exchange: the thread that owns the tile of i* hands out the
    // sibling sums of that leaf (levels inside its wave; the two levels across waves follow from the wave totals)
    {
        const int lane = tid & 63, wave = tid >> 6;
        s2[0] = wave_upsweep(s2[0], p2[0]);
        s2[1] = wave_upsweep(s2[1], p2[1]);
        if (lane == 0) {
            L.xch[0][wave] = s2[0];
            L.xch[1][wave] = s2[1];
        }
        if (tid == b_ref) {
#pragma unroll
            for (int lv = 0; lv < 6; ++lv) L.bc[lv] = p2[1].ls[lv];
        }
        __syncthreads();
        t2[0] = waves_combine(L.xch[0][0], L.xch[0][1], L.xch[0][2], L.xch[0][3], s2[0], p2[0]);
        t2[1] = waves_combine(L.xch[1][0], L.xch[1][1], L.xch[1][2], L.xch[1][3], s2[1], p2[1]);
    }
    rootW = t2[0];
    last = d.plus1 ? t2[0] + e.xw : t2[0];         // == cdf[N - 1] (the padded tree's root: left half + the last slot)
    const float Ji = fmaxf(1.0f - (d.plus1 ? t2[1] + e.xj : t2[1]), 0.0f);    // J_prob[i*] (resamplings.py:80-82)
    if (tid) L.topW[g] = make_float2(tree_left_sum(p2[0], tid), e.wf);
    // the tile that holds i*: its leaf changes from 0 to Ji, so the tile's sum is redone along the leaf's path ...
    const float tile_ref = tree_fold(Ji, e.refsib);
    // ... and so is the tree over the tiles along the path of leaf b_ref
    float bsib[8];
    {
        const int wb = b_ref >> 6;   // the wave that owns leaf b_ref
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) bsib[lv] = L.bc[lv];
        bsib[6] = L.xch[1][wb ^ 1];
        bsib[7] = (wb & 2) ? L.xch[1][0] + L.xch[1][1] : L.xch[1][2] + L.xch[1][3];
    }
    // the root of the power-of-two part's J_prob tree, and cdfJ[N - 1]
    const float rootJ = ref_extra ? t2[1] : tree_fold(tile_ref, bsib);
    const float lastJ = d.plus1 ? rootJ + (ref_extra ? Ji : e.xj) : rootJ;
    if (tid) {
        // node g: leaves [tid - 2^c, tid + 2^c), c = ctz(tid); its left half holds b_ref iff tid - 2^c <= b_ref < tid, and then
        // the half's sum is the fold of the new leaf over the first c levels of that path
        const int c = __builtin_ctz(tid);
        const bool hit = !ref_extra && b_ref < tid && b_ref >= tid - (1 << c);
        const float left = hit ? tree_fold(tile_ref, bsib, c) : tree_left_sum(p2[1], tid);
        const float jw = d.plus1 ? jprob_at(e.wf, w_max, d.N) : jprob_pow2(e.wf, w_max, inv_n);
        const float jf = tid * kBlock == i_ref ? Ji : (tid < nb ? jw : 0.0f);
        L.topJ[g] = make_float2(left, jf);
    }
    __syncthreads();
    // ---- the walk for J.  The interval of the bisection is the node itself ([lo, hi) = the node's leaves), so the
    // walk only keeps the heap index.
    const float q = lastJ * (1.0f - u3);
    const int top_levels = 31 - __builtin_clz(nb);
    float P = 0.0f, E = rootJ;
    int h = tree_walk_n(L.topJ, kBlock / nb, top_levels, q, P, E);
    const int lo0 = (h - kBlock) * kBlock;   // first slot of the tile the walk arrived at
    const float wj = d.w[lo0 + tid];
    float s1[1], t1s[1];
    TreePath p1[1];
    s1[0] = lo0 + tid == i_ref ? Ji : (d.plus1 ? jprob_at(wj, w_max, d.N) : jprob_pow2(wj, w_max, inv_n));
    const float xj = s1[0];
    block_upsweep_n<1>(s1, p1, L.xch[4], t1s);
    if (tid) L.tileJ[g] = make_float2(tree_left_sum(p1[0], tid), xj);
    __syncthreads();
    h = tree_walk_n(L.tileJ, 1, 6, q, P, E);
    {   // the node of four leaves and, in the same round trip, its children: two leaves, whose probe is E itself
        const float2 a = L.tileJ[h];
        const float4 c = *reinterpret_cast<const float4*>(L.tileJ + 2 * h);
        const bool g1 = tree_walk(a, q, P, E);
        const float tt = P + (g1 ? c.x : c.z);
        const bool gl = q <= E;
        E = gl ? tt : E;
        h = 4 * h + (g1 ? 0 : 2) + (gl ? 0 : 1);
    }
    const int leaf = lo0 + h - kBlock;
    return q <= E ? leaf : leaf + 1;
}

// Cat(w) search, LDS part: down to a node of 32 leaves (heap index h of tile `tile`)
__device__ __forceinline__ void tree_search_lds(const TreeLds& L, int nb, float q, float& P, float& E, int& tile, int& h) {
    h = tree_walk_n(L.topW, kBlock / nb, 31 - __builtin_clz(nb), q, P, E);
    tile = h - kBlock;
    h = tree_walk_n(L.midW + tile * kMidN, 1, kMidLv, q, P, E);
}

// ... the rest: nodes of 32, 16, 8 leaves from the tile's published heap in one round trip, the last four leaves of
// w in another.  K searches in lockstep (all loads of a round issued before the first is used); `on` = false: no
// loads, result untouched.
struct TreeRound {   // the nodes between the LDS levels and the last four leaves, as scalars (selects between vector
                     // loads end up in scratch): one node of 16 leaves and its children (kMidLv = 4), or three levels
    float s1, y1, s2l, y2l, s2r, y2r, s3a, y3a, s3b, y3b, s3c, y3c, s3d, y3d;
};

__device__ __forceinline__ TreeRound tree_round_load(const LgDev& d, int tile, int h) {
    const float2* tr = d.trW + (size_t)tile * kTreeNodes;
    const float2 n1 = tr[h];
    const float4 n2 = *reinterpret_cast<const float4*>(tr + 2 * h);
    if constexpr (kMidLv == 4) {
        return TreeRound{n1.x, n1.y, n2.x, n2.y, n2.z, n2.w, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    } else {
        const float4 n3a = *reinterpret_cast<const float4*>(tr + 4 * h);
        const float4 n3b = *reinterpret_cast<const float4*>(tr + 4 * h + 2);
        return TreeRound{n1.x, n1.y, n2.x, n2.y, n2.z, n2.w, n3a.x, n3a.y, n3a.z, n3a.w, n3b.x, n3b.y, n3b.z, n3b.w};
    }
}

// walks those levels; returns the first of the four leaves the walk arrived at
__device__ __forceinline__ int tree_round_walk(const TreeRound r, int tile, int h, float q, float& P, float& E) {
    static_assert(kMidLv == 3 || kMidLv == 4, "the round covers the levels between the LDS part and the last four leaves");
    const bool g1 = tree_walk(make_float2(r.s1, r.y1), q, P, E);
    const bool g2 = tree_walk(make_float2(g1 ? r.s2l : r.s2r, g1 ? r.y2l : r.y2r), q, P, E);
    int h4 = 4 * h + (g1 ? 0 : 2) + (g2 ? 0 : 1);
    if constexpr (kMidLv == 3) {
        const float s3 = g1 ? (g2 ? r.s3a : r.s3b) : (g2 ? r.s3c : r.s3d);
        const float y3 = g1 ? (g2 ? r.y3a : r.y3b) : (g2 ? r.y3c : r.y3d);
        h4 = 2 * h4 + (tree_walk(make_float2(s3, y3), q, P, E) ? 0 : 1);
    }
    return tile * kBlock + (h4 - 64) * 4;   // a node of four leaves: 64 <= h4 < 128
}

// the last four leaves (w4 = w[lo .. lo + 3]) and the closing one-leaf level
__device__ __forceinline__ int tree_leaves_walk(float4 w4, int lo, float q, float P, float E) {
    const bool g4 = tree_walk(make_float2(w4.x + w4.y, w4.z), q, P, E);
    const float tt = P + (g4 ? w4.x : w4.z);
    const bool g5 = q <= E;                                                  // two leaves: the probe is E itself
    const float e1 = g5 ? tt : E;
    const int leaf = lo + (g4 ? 0 : 2) + (g5 ? 0 : 1);
    return q <= e1 ? leaf : leaf + 1;
}

// HALVES = 2, 4: a workgroup of 512 / 1024 threads owns adjacent tiles; its first 256 threads build the trees and find J
// (the other waves wait at the barriers instead of repeating ~450 instructions each), then every thread does its slot.
template <int DMAX, int HALVES>
__global__ void __launch_bounds__(kBlock * HALVES) k_lg_prop1t(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ TreeLds L;
    __shared__ float part[2][4 * HALVES];
    __shared__ int Jsh;
    __shared__ float lastsh;
    FBSMI_STAMP(6)
    FBSMI_SPAN_IN(1, s, 4)
    const int N = d.N, tid = threadIdx.x & (kBlock - 1), half = threadIdx.x / kBlock;
    int bx = blockIdx.x;
    if (dd.pin) {   // (pinned launches use HALVES == 1)
        if (blockIdx.x & 7) return;
        bx = blockIdx.x >> 3;
    }
    const int tileb = bx * HALVES + half;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int m = tileb * kBlock + tid;   // N is a multiple of the tile: every slot is live -- but for N = 2^k + 1, whose last
    const bool live = m < N;              // workgroup holds one slot (LgDev.plus1; HALVES == 1 then)
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0: everything addressable now
    const float w_max = d.scal[1];
    const float inv_n = 1.0f / (float)N;   // N is a power of two: x / N == x * inv_n exactly
    TreeEntry te{};
    if (half == 0) te = tree_entry_loads(d, i_ref);
    float uref[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) uref[r] = r < d.du ? up[(size_t)r * N + i_ref] : 0.0f;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    float xi[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r)
        xi[r] = r < d.du ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)(live ? m : 0) * d.du + r) : 0.0f;
    FBSMI_STAMP(7)
    float last, rootW;
    int J;
    if (HALVES == 1) {
        J = tree_build(d, te, L, i_ref, u3, w_max, inv_n, last, rootW);
    } else {
        if (half == 0) {
            J = tree_build(d, te, L, i_ref, u3, w_max, inv_n, last, rootW);
            if (tid == 0) {
                Jsh = J;
                lastsh = last;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTreeBuildBarriers; ++k) __syncthreads();   // the barriers of tree_build
        }
        __syncthreads();
        J = Jsh;
        last = lastsh;
        rootW = last;    // (several tiles per workgroup: powers of two only)
    }
    int shift = (j_ref - J) % N;   // roll by j - J (:85)
    if (shift < 0) shift += N;
    int src = m - shift;
    if (src < 0) src += N;
    if (!live) src = 0;
    FBSMI_STAMP(9)
    // ---- round 2
    const float ws = d.w[src];
    float u[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + src] : 0.0f;
    const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src);
    const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)src);
    const float qK[1] = {last * (1.0f - u2)};                               // resamplings.py:73-74
    float P[1] = {0.0f}, E[1] = {rootW};
    int tile[1], h[1], hi[1] = {0};
    tree_search_lds(L, d.nb - d.plus1, qK[0], P[0], E[0], tile[0], h[0]);
    const bool killed[1] = {live && u1 * w_max >= ws};                      // :71
    FBSMI_STAMP(10)
    // ---- rounds 3, 4 (killed slots only)
    // The ancestor is one of the last four leaves or the slot behind them: for narrow states its row is fetched together
    // with those leaves (one dependent round trip less).
    constexpr bool kEarlyRow = DMAX <= 2;
    float ucand[DMAX];
    if (killed[0]) {
        const TreeRound rd = tree_round_load(d, tile[0], h[0]);
        const int lo = tree_round_walk(rd, tile[0], h[0], qK[0], P[0], E[0]);
        const float4 w4 = *reinterpret_cast<const float4*>(d.w + lo);
        float4 ug[DMAX];
        float ue[DMAX];
        if (kEarlyRow) {
            const int e = lo + 4 < N ? lo + 4 : N - 1;
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                ug[r] = r < d.du ? *reinterpret_cast<const float4*>(up + (size_t)r * N + lo) : make_float4(0.f, 0.f, 0.f, 0.f);
                ue[r] = r < d.du ? up[(size_t)r * N + e] : 0.0f;
            }
        }
        hi[0] = tree_leaves_walk(w4, lo, qK[0], P[0], E[0]);
        if (kEarlyRow) {
            const int k = hi[0] - lo;
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                const float lo2 = (k & 1) ? ug[r].y : ug[r].x, hi2 = (k & 1) ? ug[r].w : ug[r].z;
                ucand[r] = k >= 4 ? ue[r] : ((k & 2) ? hi2 : lo2);
            }
        }
    }
    const bool pinned = m == j_ref;
    const int a = pinned ? i_ref : (killed[0] ? hi[0] : src);               // :86
    FBSMI_STAMP(11)
    // ---- round 5
    if (killed[0] && !pinned) {
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            if (r < d.du) u[r] = kEarlyRow ? ucand[r] : up[(size_t)r * N + a];
    }
    if (pinned) {
#pragma unroll
        for (int r = 0; r < DMAX; ++r) u[r] = uref[r];
    }
    float lv[1];
    if (d.As && live) d.As[(size_t)s * N + m] = a;
    // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
        if (r < d.du) {
            const float dr = drift_row<DMAX>(t, r, u, v_prev);
            float x = (u[r] + dr * t.dt) + t.sd * xi[r];
            if (pinned) x = ustar[r];
            if (live) {
                un[(size_t)r * N + m] = x;
                if (d.uss) d.uss[((size_t)(s + 1) * N + m) * d.du + r] = x;
            }
        }
    }
    const float l = live ? lg_loglik<DMAX>(t, u, v, v_prev) : -__builtin_inff();   // likelihood_logpdf on the gathered particle (csmc.py:145)
    if (live) d.lw[m] = l;
    lv[0] = l;
    FBSMI_STAMP(12)
    float mx, sx;
    {   // the tile's (max, sumexp): block_lse_partial per half
        const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3, h4 = half * 4;
        const float mw = wave_max(lv[0]);
        if (lane == 0) part[0][h4 + wv] = mw;
        __syncthreads();
        mx = fmaxf(fmaxf(part[0][h4], part[0][h4 + 1]), fmaxf(part[0][h4 + 2], part[0][h4 + 3]));
        TreePath pth;
        const float sw_ = wave_upsweep(fbsmi_expf(lv[0] - finite_or_zero_f(mx)), pth);
        if (lane == 0) part[1][h4 + wv] = sw_;
        __syncthreads();
        sx = (part[1][h4] + part[1][h4 + 1]) + (part[1][h4 + 2] + part[1][h4 + 3]);
    }
    if (tid == 0) {
        d.bmax[tileb] = mx;
        d.bsumexp[tileb] = sx;
    }
    FBSMI_STAMP(13)
    FBSMI_SPAN_OUT(1, s, 4)
}

// The two-launch step with TWO slots per thread (see k_lg_prop2 below for the pairing): the trees, J and the
// staging are built once per 512 slots.
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_prop2t(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ TreeLds L;
    __shared__ float xch2[2][8];
    const int N = d.N, half = N >> 1, tid = threadIdx.x;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int mA = blockIdx.x * kBlock + tid;   // < N/2
    const int m[2] = {mA, mA + half};
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0
    const float w_max = d.scal[1];
    const float inv_n = 1.0f / (float)N;
    const TreeEntry te = tree_entry_loads(d, i_ref);
    float uref[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) uref[r] = r < d.du ? up[(size_t)r * N + i_ref] : 0.0f;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    float xi[2][DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
        xi[0][r] = 0.0f;
        xi[1][r] = 0.0f;
        if (r < d.du) {   // element mA * du + r is in the first half of the draw, its partner belongs to slot mA + N/2
            uint32_t lo_, hi_;
            random_bits_pair(t0, t1, (uint64_t)N * d.du, (uint64_t)mA * d.du + r, lo_, hi_);
            xi[0][r] = normal_from_bits(lo_);
            xi[1][r] = normal_from_bits(hi_);
        }
    }
    float last;
    float rootW_unused;
    const int J = tree_build(d, te, L, i_ref, u3, w_max, inv_n, last, rootW_unused);
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
    int src[2];
    src[0] = mA - shift;
    if (src[0] < 0) src[0] += N;
    const bool a_low = src[0] < half;          // the two sources are N/2 apart too
    src[1] = a_low ? src[0] + half : src[0] - half;
    const int pbase = a_low ? src[0] : src[1];
    // ---- round 2
    float ws[2], u[2][DMAX];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        ws[h2] = d.w[src[h2]];
#pragma unroll
        for (int r = 0; r < DMAX; ++r) u[h2][r] = r < d.du ? up[(size_t)r * N + src[h2]] : 0.0f;
    }
    uint32_t k_lo, k_hi, r_lo, r_hi;
    random_bits_pair(a0, a1, (uint64_t)N, (uint64_t)pbase, k_lo, k_hi);
    random_bits_pair(b0, b1, (uint64_t)N, (uint64_t)pbase, r_lo, r_hi);
    const float u1[2] = {fbsmi_bits_to_unit(a_low ? k_lo : k_hi), fbsmi_bits_to_unit(a_low ? k_hi : k_lo)};
    const float u2[2] = {fbsmi_bits_to_unit(a_low ? r_lo : r_hi), fbsmi_bits_to_unit(a_low ? r_hi : r_lo)};
    const float qK[2] = {last * (1.0f - u2[0]), last * (1.0f - u2[1])};    // resamplings.py:73-74
    float P[2] = {0.0f, 0.0f}, E[2] = {last, last};
    int tile[2], h[2], hi[2] = {0, 0};
    {   // the two LDS walks in lockstep (all reads of a round trip issued before the first is used), two levels per trip
        auto walk2x2 = [&](const float2* hp0, const float2* hp1, int& h0, int& h1) {
            const float2 a0 = hp0[h0], a1 = hp1[h1];
            const float4 c0 = *reinterpret_cast<const float4*>(hp0 + 2 * h0), c1 = *reinterpret_cast<const float4*>(hp1 + 2 * h1);
            const bool g0 = tree_walk(a0, qK[0], P[0], E[0]), g1 = tree_walk(a1, qK[1], P[1], E[1]);
            const bool k0 = tree_walk(g0 ? make_float2(c0.x, c0.y) : make_float2(c0.z, c0.w), qK[0], P[0], E[0]);
            const bool k1 = tree_walk(g1 ? make_float2(c1.x, c1.y) : make_float2(c1.z, c1.w), qK[1], P[1], E[1]);
            h0 = 4 * h0 + (g0 ? 0 : 2) + (k0 ? 0 : 1);
            h1 = 4 * h1 + (g1 ? 0 : 2) + (k1 ? 0 : 1);
        };
        auto walk1x2 = [&](const float2* hp0, const float2* hp1, int& h0, int& h1) {
            const float2 n0 = hp0[h0], n1 = hp1[h1];
            h0 = 2 * h0 + (tree_walk(n0, qK[0], P[0], E[0]) ? 0 : 1);
            h1 = 2 * h1 + (tree_walk(n1, qK[1], P[1], E[1]) ? 0 : 1);
        };
        const int nb = d.nb;
        int h0 = kBlock / nb, h1 = h0, lv = 31 - __builtin_clz(nb);
        for (; lv >= 2; lv -= 2) walk2x2(L.topW, L.topW, h0, h1);
        if (lv) walk1x2(L.topW, L.topW, h0, h1);
        tile[0] = h0 - kBlock;
        tile[1] = h1 - kBlock;
        const float2 *m0 = L.midW + tile[0] * kMidN, *m1 = L.midW + tile[1] * kMidN;
        h0 = h1 = 1;
        static_assert(kMidLv == 3 || kMidLv == 4, "walk of the tile levels kept in LDS");
        walk2x2(m0, m1, h0, h1);
        if (kMidLv == 4) walk2x2(m0, m1, h0, h1);
        else walk1x2(m0, m1, h0, h1);
        h[0] = h0;
        h[1] = h1;
    }
    const bool killed[2] = {u1[0] * w_max >= ws[0], u1[1] * w_max >= ws[1]};   // :71
    // ---- rounds 3, 4
    {   // both searches in lockstep: the loads of a round are issued before the first is used
        TreeRound rd0{}, rd1{};
        if (killed[0]) rd0 = tree_round_load(d, tile[0], h[0]);
        if (killed[1]) rd1 = tree_round_load(d, tile[1], h[1]);
        int lo0 = 0, lo1 = 0;
        float4 w40 = make_float4(0.f, 0.f, 0.f, 0.f), w41 = w40;
        if (killed[0]) {
            lo0 = tree_round_walk(rd0, tile[0], h[0], qK[0], P[0], E[0]);
            w40 = *reinterpret_cast<const float4*>(d.w + lo0);
        }
        if (killed[1]) {
            lo1 = tree_round_walk(rd1, tile[1], h[1], qK[1], P[1], E[1]);
            w41 = *reinterpret_cast<const float4*>(d.w + lo1);
        }
        if (killed[0]) hi[0] = tree_leaves_walk(w40, lo0, qK[0], P[0], E[0]);
        if (killed[1]) hi[1] = tree_leaves_walk(w41, lo1, qK[1], P[1], E[1]);
    }
    float lnew[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const bool pinned = m[h2] == j_ref;
        const int a = pinned ? i_ref : (killed[h2] ? hi[h2] : src[h2]);     // :86
        // ---- round 5
        if (killed[h2] && !pinned) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r)
                if (r < d.du) u[h2][r] = up[(size_t)r * N + a];
        }
        if (pinned) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[h2][r] = uref[r];
        }
        if (d.As) d.As[(size_t)s * N + m[h2]] = a;
        // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            if (r < d.du) {
                const float dr = drift_row<DMAX>(t, r, u[h2], v_prev);
                float x = (u[h2][r] + dr * t.dt) + t.sd * xi[h2][r];
                if (pinned) x = ustar[r];
                un[(size_t)r * N + m[h2]] = x;
                if (d.uss) d.uss[((size_t)(s + 1) * N + m[h2]) * d.du + r] = x;
            }
        }
        lnew[h2] = lg_loglik<DMAX>(t, u[h2], v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
        d.lw[m[h2]] = lnew[h2];
    }
    float mxA, sxA, mxB, sxB;
    block_lse_partial2(lnew[0], lnew[1], xch2[0], xch2[1], mxA, sxA, mxB, sxB);
    if (tid == 0) {
        const int tb = blockIdx.x + (d.nb >> 1);
        d.bmax[blockIdx.x] = mxA;
        d.bsumexp[blockIdx.x] = sxA;
        d.bmax[tb] = mxB;
        d.bsumexp[tb] = sxB;
    }
}

// ------------------------------------------------------------------------------------------
// The same step with TWO slots per thread, m and m + N/2 (N a multiple of 512): jax's random_bits puts
// elements i and i + n/2 of a draw on the two output words of one Threefry call, and the rotation keeps
// the two sources N/2 apart as well, so the kill-test uniform, the redraw uniform and the noise of both slots
// come out of three block-cipher calls instead of six; the heaps, J and the tables are fetched once for 512
// slots.  A workgroup then owns tiles b and b + nb/2 and publishes both logsumexp partials.
// ------------------------------------------------------------------------------------------
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_prop2(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][8];
    __shared__ __attribute__((aligned(16))) float heapW[kHeapSizeW];
    __shared__ float heapJ[kHeapSizeJ];
    __shared__ float win[kBlock];
    const int N = d.N, half = N >> 1;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3], t0 = kt[6], t1 = kt[7];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int mA = blockIdx.x * kBlock + threadIdx.x;   // < N/2
    const int m[2] = {mA, mA + half};
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    // ---- round 0
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    // the compact heaps, whole (nodes past the published depth are zero and never visited): two float4 per thread
    static_assert(kHeapSizeW == 8 * kBlock && kHeapSizeJ == kBlock, "heap staging assumes 2048 / 256 nodes");
    const float4 hw0 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x];
    const float4 hw1 = reinterpret_cast<const float4*>(d.hpW)[2 * threadIdx.x + 1];
    const float hj = d.hpJ[threadIdx.x];
    float uref[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) uref[r] = r < d.du ? up[(size_t)r * N + i_ref] : 0.0f;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    const float u3 = __uint_as_float(kt[4]);
    float xi[2][DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
        xi[0][r] = 0.0f;
        xi[1][r] = 0.0f;
        if (r < d.du) {   // element mA * du + r is in the first half of the draw, its partner belongs to slot mA + N/2
            uint32_t lo_, hi_;
            random_bits_pair(t0, t1, (uint64_t)N * d.du, (uint64_t)mA * d.du + r, lo_, hi_);
            xi[0][r] = normal_from_bits(lo_);
            xi[1][r] = normal_from_bits(hi_);
        }
    }
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x] = hw0;
    reinterpret_cast<float4*>(heapW)[2 * threadIdx.x + 1] = hw1;
    heapJ[threadIdx.x] = hj;
    __syncthreads();
    // ---- round 1: J (resamplings.py:84), the rotation j - J (:85)
    const int J = bisect_uniform(d.cdfJ, N, d.levels, d.lh_j, heapJ, win, lastJ * (1.0f - u3));
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
    int src[2];
    src[0] = mA - shift;
    if (src[0] < 0) src[0] += N;
    const bool a_low = src[0] < half;          // the two sources are N/2 apart too
    src[1] = a_low ? src[0] + half : src[0] - half;
    const int pbase = a_low ? src[0] : src[1];
    // ---- round 2
    float ws[2], u[2][DMAX];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        ws[h2] = d.w[src[h2]];
#pragma unroll
        for (int r = 0; r < DMAX; ++r) u[h2][r] = r < d.du ? up[(size_t)r * N + src[h2]] : 0.0f;
    }
    uint32_t k_lo, k_hi, r_lo, r_hi;
    random_bits_pair(a0, a1, (uint64_t)N, (uint64_t)pbase, k_lo, k_hi);
    random_bits_pair(b0, b1, (uint64_t)N, (uint64_t)pbase, r_lo, r_hi);
    const float u1[2] = {fbsmi_bits_to_unit(a_low ? k_lo : k_hi), fbsmi_bits_to_unit(a_low ? k_hi : k_lo)};
    const float u2[2] = {fbsmi_bits_to_unit(a_low ? r_lo : r_hi), fbsmi_bits_to_unit(a_low ? r_hi : r_lo)};
    const float qK[2] = {last * (1.0f - u2[0]), last * (1.0f - u2[1])};    // resamplings.py:73-74
    int lo[2], hi[2];
    bisect_lds_levels_x2(N, d.lh_w, heapW, qK, lo, hi);
    const bool killed[2] = {u1[0] * w_max >= ws[0], u1[1] * w_max >= ws[1]};   // :71
    // ---- rounds 3, 4
#pragma unroll 1
    for (int rem = d.levels - d.lh_w; rem > 0; rem -= 3) bisect_round3_x2(d.cdf, lo, hi, qK, killed);
    float lnew[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const bool pinned = m[h2] == j_ref;
        const int a = pinned ? i_ref : (killed[h2] ? hi[h2] : src[h2]);     // :86
        // ---- round 5
        if (killed[h2] && !pinned) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r)
                if (r < d.du) u[h2][r] = up[(size_t)r * N + a];
        }
        if (pinned) {
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[h2][r] = uref[r];
        }
        if (d.As) d.As[(size_t)s * N + m[h2]] = a;
        // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            if (r < d.du) {
                const float dr = drift_row<DMAX>(t, r, u[h2], v_prev);
                float x = (u[h2][r] + dr * t.dt) + t.sd * xi[h2][r];
                if (pinned) x = ustar[r];
                un[(size_t)r * N + m[h2]] = x;
                if (d.uss) d.uss[((size_t)(s + 1) * N + m[h2]) * d.du + r] = x;
            }
        }
        lnew[h2] = lg_loglik<DMAX>(t, u[h2], v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
        d.lw[m[h2]] = lnew[h2];
    }
    float mxA, sxA, mxB, sxB;
    block_lse_partial2(lnew[0], lnew[1], xch[0], xch[1], mxA, sxA, mxB, sxB);
    if (threadIdx.x == 0) {
        const int tb = blockIdx.x + (d.nb >> 1);
        d.bmax[blockIdx.x] = mxA;
        d.bsumexp[blockIdx.x] = sxA;
        d.bmax[tb] = mxB;
        d.bsumexp[tb] = sxB;
    }
}

// ------------------------------------------------------------------------------------------
// Wide models (16 < max(du, dv) <= 128, the d = 100 Gaussian-process toy of the reference's
// experiments/bashes/toy_gibbs.sh): the affine drift is a (slots x D) x (D x D) product, so it runs on
// the matrix cores.  v_mfma_f32_16x16x4_f32 accumulates as an ascending fmaf chain, bit for bit
// (tools/mfmatest.hip), which is exactly the c-ordered chain of drift_row / the oracle.
//
// One workgroup of the drift kernel = 32 destination slots x 32 rows of the drift; the grid spans (slot
// tiles x row tiles), every workgroup of a slot tile gathering that tile's ancestor rows.
// Rows < du become new particle coordinates (Euler-Maruyama + pin), rows >= du become per-row
// log-density terms lpw[m][r']; k_lgw_lse adds those in row order (the reference's sum) and
// publishes the logsumexp tile partials.  Per step: norm -> cdf -> k_lgw_anc -> k_lgw_gemm -> k_lgw_lse,
// or, for ensembles of at most 256 particles (one tile), a single launch: k_lgw_gemm<true>.
// ------------------------------------------------------------------------------------------
constexpr int kWideTile = 32;
typedef float mfma_f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kBlock) k_lgw_init(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    const int p0 = blockIdx.x * kBlock;
    // gibbs.py:140-144 (explicit_final=False): every particle starts at us_star[0], weights 1/N;
    // gibbs.py:132-138 (explicit_final=True): N(0, I) draws except the pinned slot, weights from the
    // likelihood -- those come from a drift product on these particles (the launch after this one)
    const int b0 = d.bs[0];
    for (int e = threadIdx.x; e < kBlock * d.du; e += kBlock) {
        const int p = p0 + e / d.du, r = e % d.du;
        if (p < d.N) {
            float v = d.us_star[r];
            if (d.ef && p != b0) v = normal_at(d.misc[2], d.misc[3], (uint64_t)d.N * d.du, (uint64_t)p * d.du + r);
            d.u0[(size_t)p * d.du + r] = v;
            if (d.uss) d.uss[(size_t)p * d.du + r] = v;
        }
    }
    if (d.N <= kBlock) {   // one-tile ensembles take their noise from d.xiw: step 0's is drawn here (one workgroup: N <= 256)
        wide_draw_ahead(d, 0, 0, 1);
    }
    const int p = p0 + threadIdx.x;
    float lv[1] = {-__builtin_inff()};
    if (p < d.N) {
        d.lw[p] = d.lw_init;
        lv[0] = d.lw_init;
    }
    float m, sx;
    block_lse_partial<1>(lv, xch[0], xch[1], m, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = m;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// lw[m] = lp_0 + lp_1 + ... in row order (the reference's sum over the observation coordinates).
// lpw is [N][dvp] (dvp = dv rounded up to 4): a slot's terms are at most 32 float4 loads, all issued
// before the first add, so the chain of adds waits for memory once.
struct LgwRowLoads {
    float4 x[32];
};

__device__ __forceinline__ void lgw_row_issue(const LgDev& d, int m, LgwRowLoads& L) {
    const int dvp = (d.dv + 3) & ~3;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(d.lpw + (size_t)m * dvp);
#pragma unroll
    for (int q = 0; q < 32; ++q) L.x[q] = 4 * q < d.dv ? p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float lgw_row_add(const LgDev& d, const LgwRowLoads& L) {
    float a = L.x[0].x;
    if (1 < d.dv) a = a + L.x[0].y;
    if (2 < d.dv) a = a + L.x[0].z;
    if (3 < d.dv) a = a + L.x[0].w;
#pragma unroll
    for (int q = 1; q < 32; ++q) {
        if (4 * q < d.dv) a = a + L.x[q].x;
        if (4 * q + 1 < d.dv) a = a + L.x[q].y;
        if (4 * q + 2 < d.dv) a = a + L.x[q].z;
        if (4 * q + 3 < d.dv) a = a + L.x[q].w;
    }
    return a;
}

__device__ __forceinline__ float lgw_row_sum(const LgDev& d, int m) {
    LgwRowLoads L;
    lgw_row_issue(d, m, L);
    return lgw_row_add(d, L);
}

// log-weights from the per-row terms, then the logsumexp tile partials (N > 256, and once after the
// last step for the final-mode kernels)
__global__ void __launch_bounds__(kBlock) k_lgw_lse(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    const int m = blockIdx.x * kBlock + threadIdx.x;
    float lv[1] = {-__builtin_inff()};
    if (m < d.N) {
        lv[0] = lgw_row_sum(d, m);
        d.lw[m] = lv[0];
    }
    float mx, sx;
    block_lse_partial<1>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// ancestors of one step, N > 256: the search half of k_lg_prop1 (J, rotation, kill test, Cat(w) redraw, pin)
__global__ void __launch_bounds__(kBlock) k_lgw_anc(LgDev dd, int s) {
    if (dd.nz && (int)blockIdx.x >= dd.nb) {
        wide_noise_share(chain_view(dd, blockIdx.y), s, 2, blockIdx.x - dd.nb, gridDim.x - dd.nb);
        return;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float heapW[kHeapSizeW], heapJ[kHeapSizeJ], win[kBlock];
    const int N = d.N, t = threadIdx.x;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const int m = blockIdx.x * kBlock + t;
    const bool live = m < N;
    const float lastJ = d.cdfJ[N - 1];
    const float last = d.cdf[N - 1];
    const float w_max = d.scal[1];
    constexpr int kPerThread = kHeapSizeW / kBlock;
    const int nodesW = 1 << d.lh_w, nodesJ = 1 << d.lh_j;
    float hw[kPerThread];
#pragma unroll
    for (int h = 0; h < kPerThread; ++h) {
        const int node = t + h * kBlock;
        hw[h] = (node >= 1 && node < nodesW) ? d.hpW[node] : 0.0f;
    }
    const float hj = (t >= 1 && t < nodesJ) ? d.hpJ[t] : 0.0f;
    const float u3 = __uint_as_float(kt[4]);
#pragma unroll
    for (int h = 0; h < kPerThread; ++h) heapW[t + h * kBlock] = hw[h];
    heapJ[t] = hj;
    __syncthreads();
    const int J = bisect_uniform(d.cdfJ, N, d.levels, d.lh_j, heapJ, win, lastJ * (1.0f - u3));   // resamplings.py:84
    int shift = (j_ref - J) % N;                                                                   // :85
    if (shift < 0) shift += N;
    int src = m - shift;
    if (src < 0) src += N;
    if (!live) src = 0;
    const float ws = d.w[src];
    const float u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)src);
    const float u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)src);
    const float qK = last * (1.0f - u2);                                                           // :73-74
    int lo, hi;
    bisect_lds_levels(N, d.lh_w, heapW, qK, lo, hi);
    const bool killed = live && (u1 * w_max >= ws);                                                // :71
#pragma unroll 1
    for (int rem = d.levels - d.lh_w; rem > 0; rem -= 3) bisect_round3(d.cdf, lo, hi, qK, killed);
    if (live) {
        const int a = m == j_ref ? i_ref : (killed ? hi : src);                                    // :86
        d.anc[m] = a;
        if (d.As) d.As[(size_t)s * N + m] = a;
    }
}

// N <= 256: the whole ensemble is one logsumexp tile and fits one workgroup, so nothing between two drift
// products needs a grid-wide step: log-weights (row sums), normalisation, both CDFs, J, the kill
// tests and the Cat(w) redraws run back to back inside every workgroup of the drift kernel (the
// redundancy is a few microseconds of a CU that would otherwise wait for a launch), the CDFs never
// leave LDS, and a step is ONE launch.  Same arithmetic, call for call, as
// k_lgw_lse -> k_lg_norm<1,0> -> k_lg_cdf<1,0> -> k_lgw_anc with one tile.  `store`: this workgroup
// writes the by-products (lw, w, lwn, stored paths); every workgroup gets the ancestors in ancS.
struct LgwPreLds {
    float xch[6][4];
    float cW[kBlock], cJ[kBlock];
    int ancP[kBlock];   // ancestor of SOURCE slot p (itself, or its Cat(w) redraw if killed)
    int ancS[kBlock];   // ancestor of DESTINATION slot m
};

// `early` runs between the issue of this thread's loads and their first use (the drift kernel draws its
// noise there).
template <bool ROWS, typename Early>
__device__ __forceinline__ void lgw_pre_body(const LgDev& d, int s, bool store, LgwPreLds& L, Early early) {
    const int N = d.N, t = threadIdx.x;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t a0 = kt[0], a1 = kt[1], b0 = kt[2], b1 = kt[3];
    const int i_ref = d.bs[s], j_ref = d.bs[s + 1];
    const bool live = t < N;
    FBSMI_STAMP(25)
    // log-weights: step 0 has them from k_lgw_init, later steps sum the rows the drift kernel left
    LgwRowLoads rows;
    float l = 0.0f;
    if (live) {
        if (!ROWS || (s == 0 && !d.ef)) l = d.lw[t];
        else lgw_row_issue(d, t, rows);
    }
    // in the shadow of those loads: the caller's work and the three uniforms of this thread -- as a SOURCE
    // slot p = t it owns the kill test and the redraw of p (resamplings.py:71-74), whatever the rotation.
    // (The scheduling barrier keeps the compiler from sinking the loads below this ALU work.)
    __builtin_amdgcn_sched_barrier(0);
    early();
    const float u3 = __uint_as_float(kt[4]);
    float u1 = 0.0f, u2 = 0.0f;
    if (ROWS && d.uw) {   // drawn ahead by the previous launch (wide_draw_ahead): two loads instead of two Threefry calls
        if (live) {
            u1 = d.uw[(size_t)(s & 1) * 2 * N + t];
            u2 = d.uw[(size_t)(s & 1) * 2 * N + N + t];
        }
    } else if (live) {
        u1 = uniform_at(a0, a1, (uint64_t)N, (uint64_t)t);
        u2 = uniform_at(b0, b1, (uint64_t)N, (uint64_t)t);
    }
    FBSMI_STAMP(31)
    __builtin_amdgcn_sched_barrier(0);
    if (ROWS && live && (s || d.ef)) {
        l = lgw_row_add(d, rows);
        if (store) d.lw[t] = l;
    }
    FBSMI_STAMP(26)
    // normalise (csmc.py:146).  One tile: the two-level logsumexp combine multiplies the tile's sum by
    // exp(0) = 1 and adds zeros, i.e. lse = log(sum) + max' exactly.
    float lv[1] = {live ? l : -__builtin_inff()};
    float Mraw, sx;
    block_lse_partial<1>(lv, L.xch[0], L.xch[1], Mraw, sx);
    const float lse = fbsmi_logf(sx) + finite_or_zero_f(Mraw);
    FBSMI_STAMP(27)
    const float w_max = fbsmi_expf(Mraw - lse);
    float w = 0.0f, xj = 0.0f;
    if (live) {
        const float ln = l - lse;
        w = fbsmi_expf(ln);
        if (store) {
            d.w[t] = w;
            d.lwn[t] = ln;
            if (d.lwss) d.lwss[(size_t)s * N + t] = ln;
        }
        xj = t == i_ref ? 0.0f : jprob_at(w, w_max, N);
    }
    // totals of w and of J_prob without i*; J_prob[i*] = max(1 - sum, 0) (resamplings.py:80-82)
    float s2[2] = {w, xj}, t2[2];
    TreePath p2[2];
    block_upsweep_n<2>(s2, p2, L.xch[2], t2);
    const float Ji = fmaxf(1.0f - t2[1], 0.0f);
    const float xo = live ? (t == i_ref ? Ji : xj) : 0.0f;
    float s1[1] = {xo}, t1[1];
    TreePath p1[1];
    block_upsweep_n<1>(s1, p1, L.xch[4], t1);
    // canonical cumsums of the single tile: (P, E) = (0, total)
    {
        float P = 0.0f, E = t2[0], c[1];
        const float xw1[1] = {w};
        block_descend(P, E, p2[0]);
        chunk_scan<1>(xw1, P, E, c);
        L.cW[t] = c[0];
        P = 0.0f;
        E = t1[0];
        const float xo1[1] = {xo};
        block_descend(P, E, p1[0]);
        chunk_scan<1>(xo1, P, E, c);
        L.cJ[t] = c[0];
    }
    __syncthreads();
    FBSMI_STAMP(28)
    // conditional killing (resamplings.py:66-86) on the LDS-resident CDFs: J and the per-source redraws are
    // independent searches, their LDS walks overlap
    const int J = searchsorted_left(L.cJ, N, d.levels, L.cJ[N - 1] * (1.0f - u3));
    int ap = t;
    if (live && u1 * w_max >= w) ap = searchsorted_left(L.cW, N, d.levels, L.cW[N - 1] * (1.0f - u2));
    L.ancP[t] = ap;
    int shift = (j_ref - J) % N;
    if (shift < 0) shift += N;
    __syncthreads();
    int a = -1;
    if (live) {
        int src = t - shift;
        if (src < 0) src += N;
        a = t == j_ref ? i_ref : L.ancP[src];
        if (store && d.As) d.As[(size_t)s * N + t] = a;
    }
    L.ancS[t] = a;
    __syncthreads();
    FBSMI_STAMP(29)
}

// The same for the particle filters (bootstrap_filter smc.py:58-74, pmcmc_filter_step smc.py:138-152): row
// sums -> c = logsumexp -> log-likelihood accumulator -> w -> cumsum -> stratified / systematic ancestors
// (resampling.py:43-51) with the resampling key of step `kres`.  Call for call the arithmetic of
// k_lgw_lse -> k_filt_norm<1> -> k_lg_cdf<1,2> -> the search of k_filt_prop with one tile.
template <bool ROWS, typename Early>
__device__ __forceinline__ void lgw_fpre_body(const LgDev& d, int kres, bool store, LgwPreLds& L, Early early) {
    const int N = d.N, t = threadIdx.x;
    const uint32_t r0 = d.keytab[8 * kres + 2], r1 = d.keytab[8 * kres + 3];
    const bool live = t < N;
    LgwRowLoads rows;
    float l = 0.0f;
    if (live) {
        if (ROWS) lgw_row_issue(d, t, rows);
        else l = d.lw[t];
    }
    early();
    float uu = 0.0f;
    if (d.systematic) uu = uniform_at(r0, r1, 1, 0);
    else if (live) uu = uniform_at(r0, r1, (uint64_t)N, (uint64_t)t);
    if (ROWS && live) {
        l = lgw_row_add(d, rows);
        if (store) d.lw[t] = l;
    }
    float lv[1] = {live ? l : -__builtin_inff()};
    float Mraw, sx;
    block_lse_partial<1>(lv, L.xch[0], L.xch[1], Mraw, sx);
    const float c = fbsmi_logf(sx) + finite_or_zero_f(Mraw);
    if (store && t == 0) {
        const float e0 = *d.ell;
        *d.ell = d.flow == 0 ? e0 - (c - d.logn)      // log_nell -= _c - log(N)          smc.py:67
                             : (e0 - d.logn) + c;      // log_ell = log_ell - log(N) + _c  smc.py:146
    }
    const float w = live ? fbsmi_expf(l - c) : 0.0f;   // smc.py:68-69 / :147-148
    if (store && live) d.w[t] = w;
    float s1[1] = {w}, t1[1];
    TreePath p1[1];
    block_upsweep_n<1>(s1, p1, L.xch[2], t1);
    {
        float P = 0.0f, E = t1[0], cc[1];
        const float xw1[1] = {w};
        block_descend(P, E, p1[0]);
        chunk_scan<1>(xw1, P, E, cc);
        L.cW[t] = cc[0];
    }
    __syncthreads();
    int a = -1;
    if (live) {
        const float q = ((float)t + uu) / (float)N;
        a = searchsorted_left(L.cW, N, d.levels, q);
        a = a < 0 ? 0 : (a > N - 1 ? N - 1 : a);
    }
    L.ancS[t] = a;
    __syncthreads();
}

// The drift product and what hangs on it.  nrt = row tiles = ceil(D / 32); Kp = D rounded up to a multiple of
// 16; the two LDS tiles (32 rows of G_s, 32 gathered slots of z) are laid out by wide_pos (below), S floats per plane row.
// KIND: where the ancestors come from -- 0: d.anc (k_lgw_anc); 1: the Gibbs step prologue, in this
// workgroup; 2: the filter prologue (resampling key of step kres), in this workgroup; 3: identity (no
// resampling in front of this product); 4: d.anc again, with the filters' conventions (k_lgwf_anc).  tr0 / nrt: the row tiles of this launch; emit bit 0: rows < du
// are written (new particles), bit 1: rows >= du are written (log-density terms).
// LDS tiles of the drift kernels: [4 column planes][32 rows][S] floats, element (row i, column c) at ((c % 4) * 32 + i) * S + c / 4
// -- lane group g = lane / 16 of an MFMA (which supplies column 4q + g of instruction q) finds its columns of four consecutive
// instructions in one aligned float4 of plane g.  S = Kp / 4 (rounded up to 4 mod 8, so that S / 4 is odd).  A ds_read_b128 is
// served in four groups of sixteen lanes -- NOT lanes 0-15, 16-31, ... but {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35,
// 44-47, 52-59}, {36-43, 48-51, 60-63}: every group holds each of the sixteen rows (lane % 16) once, from two planes.  A plane is
// 32 S floats = a multiple of 64 banks, so planes do not shift the banks, and row r starts at bank quad (S / 4) r mod 16: the
// sixteen rows of a group cover the sixteen quads, conflict-free.  (The rows-outside layout [32][4 planes] this replaces put plane g
// 13 g quads further on: two-way conflicts in every group, 37 % of the LDS cycles of the large-ensemble kernel -- and needed
// 54 272 bytes for its two tiles at D = 200, of which a CU holds TWO, whatever the occupancy query says: tools/occ_probe.  The
// planes-outside form needs 53 248, and three workgroups per CU fit: 626 workgroups at 10 000 particles x 2 chains run in one
// round instead of two.)
__device__ __forceinline__ int wide_pos(int i, int c, int S) { return ((c & 3) * kWideTile + i) * S + (c >> 2); }
static inline int wide_plane_row(int Kp) { return ((Kp / 4) & 4) ? Kp / 4 : Kp / 4 + 4; }   // S: Kp / 4 made 4 mod 8
static inline size_t wide_lds_bytes(int S) { return sizeof(float) * 2 * 4 * kWideTile * (size_t)S; }

template <int KIND>
__global__ void __launch_bounds__(kBlock) k_lgw_gemm(LgDev dd, int s, int tr0, int nrt, int Kp, int S, int emit, int kres) {
    constexpr bool FUSED = KIND == 1;
    constexpr bool FILT = KIND >= 2;
    // KIND 1 with kres >= 8 (kres is otherwise unused there): the grid is eight times as wide and only every eighth block
    // works -- blocks b and b + 8 share an XCD, so the whole small ensemble runs on ONE XCD and what a step hands to the next
    // (log-density terms, particles, noise) is found in that XCD's L2 instead of behind the fabric.
    // (the filters' launches, whose kres is a step index, ask for it with bit 8 of `emit`: class 0)
    const bool pin = (KIND == 1 && kres >= 8) || (KIND != 1 && (emit & 0x100));
    const int pslot = KIND == 1 ? (kres - 8 + 2 * (int)blockIdx.y) & 7 : 0;   // chain y of the launch on XCD class base + 2 y
    const int bx = pin ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, gx = pin ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    if (pin && (int)(blockIdx.x & 7) != pslot) {
        // The seven blocks out of eight that sit on the other XCDs: in the Gibbs step they draw the NEXT step's noise (into the
        // other half of d.xiw) while the working blocks run -- it used to be the working blocks' last 1.4 us.
        if (KIND == 1) {
            const LgDev dq = chain_view(dd, blockIdx.y);
            if (s + 1 < dq.T) {
                const int r8 = (int)(blockIdx.x & 7), idle = bx * 7 + (r8 > pslot ? r8 - 1 : r8), nidle = gx * 7;
                wide_draw_ahead(dq, s + 1, idle, nidle);
            }
        }
        return;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    // Unpinned one-tile Gibbs launches carry a few EXTRA blocks behind the working ones: they draw the next step's noise (into
    // the other half of d.xiw) on CUs of their own while the step runs, instead of every working block spending its last
    // 1.4 us on a share of it.
    const int work = ((d.N + kWideTile - 1) / kWideTile) * nrt;   // working blocks of a launch that covers all row tiles
    const bool extra_noise = KIND == 1 && !pin && (int)gridDim.x > work;
    if (extra_noise && bx >= work) {
        if (s + 1 < d.T) wide_draw_ahead(d, s + 1, bx - work, (int)gridDim.x - work);
        return;
    }
    __shared__ LgwPreLds pre;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* Gs = dyn;                    // [32 rows][S]: rows 32*tr .. of G_s
    float* Zs = dyn + 4 * kWideTile * S;    // [32 slots][S]: z = (u[ancestor], v_prev)
    const int N = d.N, du = d.du, D = d.D;
    const int ts = bx / nrt, tr = tr0 + (bx - ts * nrt);
    // (the wave index as a SCALAR: row numbers, LDS rows and the half-tile tests that hang on it then live on the scalar unit)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t t0 = kt[FILT ? 0 : 6], t1 = kt[FILT ? 1 : 7];   // key_transition (Gibbs) / key_proposal (filters)
    const int j_ref = FILT ? -1 : d.bs[s + 1];
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    const float* __restrict__ G = d.G + (size_t)s * D * D;
    const float* __restrict__ g = d.g + (size_t)s * D;
    const float sd = d.sd[s], lognorm = d.lognorm[s];
    const float sd2 = sd * sd;
    // (KIND 3 with kres = 1: the initial weights of explicit_final, likelihood_logpdf(vs[0], u0s, vs[1], ts[0]) --
    // the observation pair the other way round, gibbs.py:136-137)
    const bool swap_v = KIND == 3 && kres == 1;
    const float* v_prev = d.vs + (size_t)(swap_v ? s + 1 : s) * d.dv;
    const float* v = d.vs + (size_t)(swap_v ? s : s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * du;
    FBSMI_STAMP(20)
    // ---- round 0.  Wave w stages rows / slots w, w+4, ... of the two tiles.  The ancestors are asked for
    //      first, the G tile next; the four noise draws of this thread run while both are in flight; the
    //      ancestor rows (the only dependent loads) go out as soon as the ancestors are here.
    constexpr int kRows = kWideTile / kWaves;   // 8 rows / slots per wave
    int an[kRows];
    if (KIND == 0 || KIND == 3 || KIND == 4) {
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj) {
            const int mj = kWideTile * ts + wave + kWaves * jj;
            an[jj] = mj < N ? (KIND == 3 ? mj : d.anc[mj]) : -1;
        }
    }
    const bool vec4 = (D & 3) == 0 && (du & 3) == 0;   // rows are whole float4s: one load per lane and row
    float gq[kRows * 4], zq[kRows * 4];
    if (vec4) {
        const int c = 4 * lane;
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj) {
            const int r = kWideTile * tr + wave + kWaves * jj;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < D && c < D) x = *reinterpret_cast<const float4*>(G + (size_t)r * D + c);
            gq[jj * 4 + 0] = x.x; gq[jj * 4 + 1] = x.y; gq[jj * 4 + 2] = x.z; gq[jj * 4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int r = kWideTile * tr + wave + kWaves * (q >> 2), c = lane + 64 * (q & 3);
            gq[q] = (r < D && c < D) ? G[(size_t)r * D + c] : 0.0f;
        }
    }
    // accumulator geometry of v_mfma_f32_16x16x4_f32: wave = (row half ar, slot half ac); register v of
    // a lane holds (row 4*(lane/16) + v, slot lane%16) of the 16 x 16 block
    const int ar = wave >> 1, ac = wave & 1;
    const int row0 = kWideTile * tr + 16 * ar + 4 * (lane >> 4);
    const int jloc = 16 * ac + (lane & 15);
    const int mo = kWideTile * ts + jloc;   // the slot this lane's outputs belong to
    mfma_f4 acc;
    float xi[4];
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) acc[vv] = row0 + vv < D ? g[row0 + vv] : 0.0f;
    // Four independent Threefry + erf_inv chains in one basic block (the compiler interleaves them: a lone
    // dependent chain issues one instruction every ~7 clocks); skipped as a whole by row tiles without rows < du.
    auto draw_noise = [&]() {
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) xi[vv] = 0.0f;
        if ((emit & 1) && kWideTile * tr < du) {
            uint32_t bits[4];
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const int r = row0 + vv;
                const bool ok = r < du && mo < N;
                bits[vv] = random_bits_at(t0, t1, (uint64_t)N * du, ok ? (uint64_t)mo * du + r : 0ull);
            }
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const float z = normal_from_bits(bits[vv]);
                xi[vv] = (row0 + vv < du && mo < N) ? z : 0.0f;
            }
        }
    };
    if (KIND == 0 || KIND == 3 || KIND == 4) draw_noise();
    // One-tile Gibbs: this step's noise was drawn by the PREVIOUS launch (every workgroup a share, at its end, where
    // the workgroups without rows < du would otherwise idle): ~1600 VALU instructions off the critical path.
    auto fetch_noise = [&]() {
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            const int r = row0 + vv;
            xi[vv] = (r < du && mo < N) ? d.xiw[(size_t)(s & 1) * N * du + (size_t)mo * du + r] : 0.0f;
        }
    };
    if (KIND == 1 || KIND == 2) {   // the G tile is on its way; now the step's ancestors, worked out by this workgroup itself
        if (KIND == 1) lgw_pre_body<true>(d, s, bx == 0, pre, fetch_noise);
        else lgw_fpre_body<true>(d, kres, bx == 0, pre, draw_noise);
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj) {
            const int mj = kWideTile * ts + wave + kWaves * jj;
            an[jj] = mj < N ? pre.ancS[mj] : -1;
        }
    }
    if (vec4) {
        const int c = 4 * lane;
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj) {
            const int a = an[jj];
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a >= 0 && c < D)
                x = c < du ? *reinterpret_cast<const float4*>(up + (size_t)a * du + c)
                           : *reinterpret_cast<const float4*>(v_prev + (c - du));
            zq[jj * 4 + 0] = x.x; zq[jj * 4 + 1] = x.y; zq[jj * 4 + 2] = x.z; zq[jj * 4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int a = an[q >> 2], c = lane + 64 * (q & 3);
            float z = 0.0f;
            if (a >= 0 && c < D) z = c < du ? up[(size_t)a * du + c] : v_prev[c - du];
            zq[q] = z;
        }
    }
    if (FILT && d.uss && d.flow == 0 && s > 0 && tr == tr0) {   // filtering_samples[s] = the resampled particles (smc.py:72,84)
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int mj = kWideTile * ts + wave + kWaves * (q >> 2);
            const int c = vec4 ? 4 * lane + (q & 3) : lane + 64 * (q & 3);
            if (mj < N && c < du) d.uss[((size_t)s * N + mj) * du + c] = zq[q];
        }
    }
    FBSMI_STAMP(21)
    const int Q = Kp >> 2;
    if (vec4) {
        // a lane's four columns 4 lane .. 4 lane + 3 lie inside Kp (a multiple of 16) together: ONE branch for the 64 stores (they
        // used to be 64 predicated stores: an exec-mask save, a branch and a join each).  The G tile, which has been here since
        // the prologue, goes first: its stores run while the ancestor rows are still on their way.
        if (4 * lane < Kp) {
#pragma unroll
            for (int q = 0; q < kRows * 4; ++q) Gs[wide_pos(wave + kWaves * (q >> 2), q & 3, S) + lane] = gq[q];
#pragma unroll
            for (int q = 0; q < kRows * 4; ++q) Zs[wide_pos(wave + kWaves * (q >> 2), q & 3, S) + lane] = zq[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int i = wave + kWaves * (q >> 2);
            const int c = lane + 64 * (q & 3);
            if (c < Kp) {
                const int pos = wide_pos(i, c, S);
                Gs[pos] = gq[q];
                Zs[pos] = zq[q];
            }
        }
    }
    __syncthreads();
    FBSMI_STAMP(22)
    // ---- drift rows: acc = g_r, then acc = fma(G[r][c], z[c], acc) for c = 0 .. D-1, on the matrix cores
    {
        const float4* ga = reinterpret_cast<const float4*>(Gs + wide_pos(16 * ar + (lane & 15), lane >> 4, S));
        const float4* zb = reinterpret_cast<const float4*>(Zs + wide_pos(jloc, lane >> 4, S));
        // two operand sets in turn: the reads of the next four products are in flight under the current four (k_lgw_gemm_fat);
        // waves whose sixteen rows lie wholly outside the matrix have nothing to multiply
        const int nq = kWideTile * tr + 16 * ar < D ? Q >> 2 : 0;
        float4 a0 = ga[0], b0 = zb[0];
#pragma unroll 1
        for (int q4 = 0; q4 < nq; q4 += 2) {
            const int q1 = q4 + 1 < nq ? q4 + 1 : q4, q2 = q4 + 2 < nq ? q4 + 2 : q4;
            const float4 a1 = ga[q1], b1 = zb[q1];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q4 + 1 < nq) {
                a0 = ga[q2];
                b0 = zb[q2];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    FBSMI_STAMP(23)
    // ---- rows < du: transition_sampler (gp_gibbs.py:120-122) + pin (csmc.py:143);
    //      rows >= du: the terms of likelihood_logpdf (gp_gibbs.py:131-135, csmc.py:145)
    if (mo < N && vec4) {
        // the lane's four rows are consecutive and du, D are multiples of four: all four are coordinates, or all four are
        // observation rows, or none exists -- and each kind leaves as ONE 16-byte store
        const bool pinned = mo == j_ref;
        if (row0 < du) {
            if (emit & 1) {
                float x[4];
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) {
                    const int r = row0 + vv;
                    x[vv] = (Zs[wide_pos(jloc, r, S)] + acc[vv] * d.dt) + sd * xi[vv];
                    if (pinned) x[vv] = ustar[r];
                }
                const float4 o4 = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(un + (size_t)mo * du + row0) = o4;
                if (!FILT && d.uss) *reinterpret_cast<float4*>(d.uss + ((size_t)(s + 1) * N + mo) * du + row0) = o4;
                if (FILT && d.flow == 1 && s == d.T - 1) *reinterpret_cast<float4*>(d.usT + (size_t)mo * du + row0) = o4;
            }
        } else if (row0 < D && (emit & 2)) {
            const int rv0 = row0 - du;
            const float4 vt = *reinterpret_cast<const float4*>(v + rv0), vp = *reinterpret_cast<const float4*>(v_prev + rv0);
            const float tg[4] = {vt.x, vt.y, vt.z, vt.w}, pv[4] = {vp.x, vp.y, vp.z, vp.w};
            float l4[4];
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) l4[vv] = norm_logpdf(tg[vv], pv[vv] + acc[vv] * d.dt, sd2, lognorm);
            *reinterpret_cast<float4*>(d.lpw + (size_t)mo * ((d.dv + 3) & ~3) + rv0) = make_float4(l4[0], l4[1], l4[2], l4[3]);
        }
    } else if (mo < N) {
        const bool pinned = mo == j_ref;
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            const int r = row0 + vv;
            if (r < du) {
                if (emit & 1) {
                    float x = (Zs[wide_pos(jloc, r, S)] + acc[vv] * d.dt) + sd * xi[vv];
                    if (pinned) x = ustar[r];
                    un[(size_t)mo * du + r] = x;
                    if (!FILT && d.uss) d.uss[((size_t)(s + 1) * N + mo) * du + r] = x;
                    if (FILT && d.flow == 1 && s == d.T - 1) d.usT[(size_t)mo * du + r] = x;
                }
            } else if (r < D && (emit & 2)) {
                const int rv = r - du;
                const float cond_m = v_prev[rv] + acc[vv] * d.dt;
                d.lpw[(size_t)mo * ((d.dv + 3) & ~3) + rv] = norm_logpdf(v[rv], cond_m, sd2, lognorm);
            }
        }
    }
    FBSMI_STAMP(30)
#ifdef FBSMI_STAMPS
    const bool stamp_tail = true;   // diagnostic build: time the tail on the last step too (its draws are never used)
#else
    const bool stamp_tail = false;
#endif
    if (KIND == 1 && !pin && !extra_noise && (s + 1 < d.T || stamp_tail)) {   // this workgroup's share of the next step's noise (pinned
        // launches: the idle blocks on the other XCDs drew it).  Into the other half: blocks of this launch may still be reading
        // this step's.  (Diagnostic build, last step: the draws of a step that does not exist, with step s's keys, timed only.)
        if (s + 1 < d.T) wide_draw_ahead(d, s + 1, bx, gx);
        else {
            LgDev dx = d;
            dx.keytab = d.keytab - 8;
            wide_draw_ahead(dx, s + 1, bx, gx);
        }
    }
    FBSMI_STAMP(24)
}

// The noise of one step of a large wide ensemble, normal(key_transition, (N, du)) (gp_gibbs.py:122), as its own streaming
// launch: elements a and a + n/2 of the draw are the two words of one Threefry call (jax's random_bits), so a thread that
// owns a pair draws two normals per block-cipher call -- half the instructions of normal_at per element inside the drift
// kernel, where four of the seven row tiles of every workgroup spent 2.5 us each on their draws with the matrix cores idle.
__global__ void __launch_bounds__(kBlock) k_lgw_noise(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const uint32_t t0 = d.keytab[8 * s + 6], t1 = d.keytab[8 * s + 7];
    const uint32_t n = (uint32_t)d.N * (uint32_t)d.du, half = (n + 1u) >> 1;
    for (uint32_t a = blockIdx.x * kBlock + threadIdx.x; a < half; a += gridDim.x * kBlock) {
        const uint32_t b = a + half;
        uint32_t o0, o1;
        threefry2x32(t0, t1, a, b < n ? b : 0u, o0, o1);
        d.xiw[a] = normal_from_bits(o0);
        if (b < n) d.xiw[b] = normal_from_bits(o1);
    }
}

// The same product for LARGE wide ensembles: one workgroup keeps its 32 gathered ancestor rows in LDS and
// walks ALL row tiles of G_s over them (the gather -- the scattered part -- is paid once instead of once
// per row tile), the next G tile travelling to registers while the matrix cores work on the current one.
// Ancestors from k_lgw_anc, all rows emitted.  66 KB of LDS: two workgroups per CU, so one's noise draws
// (VALU) overlap the other's MFMAs.
// (Measured, round 3, 10 000 particles x 2 chains = 626 workgroups, per-workgroup entry / exit stamps of the diagnostic build:
// all enter within 1.1 us; the CUs that hold three of them finish after 33 us, those with two after 22 -- 11 us of CU time per
// workgroup either way, i.e. the CU is saturated by two: per row tile a wave issues ~290 vector instructions besides its 52
// dependent products, and the two kinds overlap little.  Starting the workgroups of a CU out of phase (s_sleep by b / 256) made
// it slower.)
template <bool VEC4>   // D and du multiples of four: rows are whole float4s (the usual case; the other is kept for odd sizes)
__global__ void __launch_bounds__(kBlock) k_lgw_gemm_fat(LgDev dd, int s, int nrt, int Kp, int S) {
    const LgDev d = chain_view(dd, blockIdx.y);
#ifdef FBSMI_STAMPS
    // diagnostic build: every workgroup of the last step's launch records its entry / exit time (view 9; the second noise slot of
    // chain 0, which this path does not use) -- how the launch's workgroups are spread over its duration
    unsigned long long* span = reinterpret_cast<unsigned long long*>(dd.xiw + (size_t)dd.N * dd.du) + 2 * (blockIdx.x + gridDim.x * blockIdx.y);
    if (threadIdx.x == 0 && s == d.T - 1) span[0] = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* Gs = dyn;
    float* Zs = dyn + 4 * kWideTile * S;
    const int N = d.N, du = d.du, D = d.D;
    const int ts = blockIdx.x;
    // (the wave index as a SCALAR: row numbers, LDS rows and the half-tile tests that hang on it then live on the scalar unit)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j_ref = d.bs[s + 1];
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    const float* __restrict__ G = d.G + (size_t)s * D * D;
    const float* __restrict__ g = d.g + (size_t)s * D;
    const float sd = d.sd[s], lognorm = d.lognorm[s];
    const float sd2 = sd * sd;
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * du;
    FBSMI_STAMP(20)
    constexpr int kRows = kWideTile / kWaves;
    constexpr bool vec4 = VEC4;
    const int Q = Kp >> 2;
    int an[kRows];
#pragma unroll
    for (int jj = 0; jj < kRows; ++jj) {
        const int mj = kWideTile * ts + wave + kWaves * jj;
        an[jj] = mj < N ? d.anc[mj] : -1;
    }
    float gq[kRows * 4], zq[kRows * 4];
    auto load_g = [&](int tr) {
        if (vec4) {
            // branch-free AND select-free: every lane loads from a clamped, always valid address.  (A predicated load costs an
            // exec-mask save, a branch and a join per row; a select on the loaded value makes the wave wait for the prefetch
            // before the products instead of after them -- measured 11 % slower.)  What arrives from outside the matrix is
            // finite model data and never shows: rows >= D of the tile are not emitted, and columns D .. Kp-1 meet the exact
            // zeros of the Z tile's padding -- fma(x, 0, acc) = acc, as fma(0, 0, acc) was.
            const int c = 4 * lane < D ? 4 * lane : 0;
#pragma unroll
            for (int jj = 0; jj < kRows; ++jj) {
                const int r = kWideTile * tr + wave + kWaves * jj;
                const float4 x = *reinterpret_cast<const float4*>(G + (size_t)(r < D ? r : D - 1) * D + c);
                gq[jj * 4 + 0] = x.x; gq[jj * 4 + 1] = x.y; gq[jj * 4 + 2] = x.z; gq[jj * 4 + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kRows * 4; ++q) {
                const int r = kWideTile * tr + wave + kWaves * (q >> 2), c = lane + 64 * (q & 3);
                gq[q] = (r < D && c < D) ? G[(size_t)r * D + c] : 0.0f;
            }
        }
    };
    auto store_tile = [&](float* dst, const float (&src)[kRows * 4]) {
        if (vec4) {   // a lane's four columns 4 lane .. 4 lane + 3 are inside Kp (a multiple of 16) together: ONE branch
            if (4 * lane < Kp) {
#pragma unroll
                for (int q = 0; q < kRows * 4; ++q)
                    dst[wide_pos(wave + kWaves * (q >> 2), q & 3, S) + lane] = src[q];
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int i = wave + kWaves * (q >> 2);
            const int c = lane + 64 * (q & 3);
            if (c < Kp) dst[wide_pos(i, c, S)] = src[q];
        }
    };
    load_g(0);
    if (vec4) {
        const int c = 4 * lane;
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj) {
            const int a = an[jj];
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a >= 0 && c < D)
                x = c < du ? *reinterpret_cast<const float4*>(up + (size_t)a * du + c)
                           : *reinterpret_cast<const float4*>(v_prev + (c - du));
            zq[jj * 4 + 0] = x.x; zq[jj * 4 + 1] = x.y; zq[jj * 4 + 2] = x.z; zq[jj * 4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kRows * 4; ++q) {
            const int a = an[q >> 2], c = lane + 64 * (q & 3);
            float z = 0.0f;
            if (a >= 0 && c < D) z = c < du ? up[(size_t)a * du + c] : v_prev[c - du];
            zq[q] = z;
        }
    }
    store_tile(Gs, gq);
    store_tile(Zs, zq);
    __syncthreads();
    FBSMI_STAMP(21)
    const int ar = wave >> 1, ac = wave & 1;
    const int jloc = 16 * ac + (lane & 15);
    const int mo = kWideTile * ts + jloc;
    const bool pinned = mo == j_ref;
    const float4* ga = reinterpret_cast<const float4*>(Gs + wide_pos(16 * ar + (lane & 15), lane >> 4, S));
    const float4* zb = reinterpret_cast<const float4*>(Zs + wide_pos(jloc, lane >> 4, S));
    const int dvp = (d.dv + 3) & ~3;
    // Vector-memory loads return in order, so a wait for the LAST load issued waits for all of them: the accumulators' initial
    // values g[r] used to be loaded after the next G tile had been asked for, and the products started only when that whole
    // prefetch had arrived (half of the kernel's wave-cycles were such waits).  Now the bias of tile tr + 1 is fetched one tile
    // ahead, and within a tile the loads go out in the order of their use: next bias, this tile's noise, next G tile.
    // float4 path: a lane keeps the log-density terms of its rows >= du (four per row tile, at most five such tiles: dv <= 128)
    // in registers until the tiles are done with the LDS; they never travel through global memory (below)
    constexpr int kVTiles = 5;
    const int vt0 = du / kWideTile;   // first row tile that holds rows >= du
    float vsv[kVTiles][4];
#pragma unroll
    for (int j = 0; j < kVTiles; ++j)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) vsv[j][vv] = 0.0f;
    float gb[4];
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) {
        const int r = 16 * ar + 4 * (lane >> 4) + vv;
        gb[vv] = g[r < D ? r : D - 1];   // (rows >= D are never emitted: any finite value will do)
    }
#pragma unroll 1
    for (int tr = 0; tr < nrt; ++tr) {
        const int row0 = kWideTile * tr + 16 * ar + 4 * (lane >> 4);
        float gbn[4];
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            const int r = row0 + kWideTile + vv;
            gbn[vv] = g[r < D ? r : D - 1];
        }
        mfma_f4 acc;
        float xi[4];
        if (vec4) {   // the step's noise: one 16-byte load, in flight under the products
            // (clamped address as in load_g; the draws are only used by the lanes that own coordinates of a live slot)
            const float4 x4 = *reinterpret_cast<const float4*>(d.xiw + ((row0 < du && mo < N) ? (size_t)mo * du + row0 : (size_t)0));
            xi[0] = x4.x; xi[1] = x4.y; xi[2] = x4.z; xi[3] = x4.w;
        } else {
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) xi[vv] = (row0 + vv < du && mo < N) ? d.xiw[(size_t)mo * du + row0 + vv] : 0.0f;
        }
        if (tr + 1 < nrt) load_g(tr + 1);   // in flight while this tile is multiplied
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) acc[vv] = gb[vv];
        // the operands of the NEXT four products travel from LDS while the current four (a dependent chain, 4 x 8 passes) run:
        // two operand sets in turn, the scheduling barriers keep each read in front of the products it overlaps (left alone the
        // compiler rotates the read to the top of the next iteration and waits for it there)
        if (kWideTile * tr + 16 * ar < D) {   // (the upper half of the last row tile can lie wholly outside the matrix: D = 200 ends
            const int nq = Q >> 2;            // at row 8 of tile 6, whose waves with ar = 1 have nothing to multiply)
            __builtin_amdgcn_s_setprio(3);   // the product loop ahead of other waves' staging / epilogue code
            float4 a0 = ga[0], b0 = zb[0];
#pragma unroll 1
            for (int q4 = 0; q4 < nq; q4 += 2) {
                const int q1 = q4 + 1 < nq ? q4 + 1 : q4, q2 = q4 + 2 < nq ? q4 + 2 : q4;
                const float4 a1 = ga[q1], b1 = zb[q1];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q4 + 1 < nq) {
                    a0 = ga[q2];
                    b0 = zb[q2];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (tr == 0) { FBSMI_STAMP(22) }
        if (tr == 5) { FBSMI_STAMP(26) }
        if (mo < N && vec4) {
            // the lane's four rows are consecutive and du, D are multiples of four: all four are coordinates, or all four
            // are observation rows, or none exists -- and each kind leaves as ONE 16-byte store (they were four dword
            // stores to sixteen different cache lines per instruction)
            if (row0 < du) {
                float x[4];
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) {
                    const int r = row0 + vv;
                    x[vv] = (Zs[wide_pos(jloc, r, S)] + acc[vv] * d.dt) + sd * xi[vv];
                    if (pinned) x[vv] = ustar[r];
                }
                const float4 o4 = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(un + (size_t)mo * du + row0) = o4;
                if (d.uss) *reinterpret_cast<float4*>(d.uss + ((size_t)(s + 1) * N + mo) * du + row0) = o4;
            } else if (row0 < D) {
                const int rv0 = row0 - du;
                const float4 vt = *reinterpret_cast<const float4*>(v + rv0), vp = *reinterpret_cast<const float4*>(v_prev + rv0);
                const float tg[4] = {vt.x, vt.y, vt.z, vt.w}, pv[4] = {vp.x, vp.y, vp.z, vp.w};
                float l4[4];
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) l4[vv] = norm_logpdf(tg[vv], pv[vv] + acc[vv] * d.dt, sd2, lognorm);
                const int jt = tr - vt0;   // (uniform: a scalar branch per candidate, no indexed registers)
#pragma unroll
                for (int j = 0; j < kVTiles; ++j)
                    if (jt == j) {
#pragma unroll
                        for (int vv = 0; vv < 4; ++vv) vsv[j][vv] = l4[vv];
                    }
            }
        } else if (mo < N) {
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const int r = row0 + vv;
                if (r < du) {
                    float x = (Zs[wide_pos(jloc, r, S)] + acc[vv] * d.dt) + sd * xi[vv];
                    if (pinned) x = ustar[r];
                    un[(size_t)mo * du + r] = x;
                    if (d.uss) d.uss[((size_t)(s + 1) * N + mo) * du + r] = x;
                } else if (r < D) {
                    const int rv = r - du;
                    const float cond_m = v_prev[rv] + acc[vv] * d.dt;
                    d.lpw[(size_t)mo * dvp + rv] = norm_logpdf(v[rv], cond_m, sd2, lognorm);
                }
            }
        }
        if (tr == 0) { FBSMI_STAMP(23) }
        if (tr == 5) { FBSMI_STAMP(27) }
        if (tr + 1 < nrt) {
            __syncthreads();      // every wave is done reading the G tile
            store_tile(Gs, gq);
            __syncthreads();
        }
        if (tr == 0) { FBSMI_STAMP(24) }
        if (tr == 4) { FBSMI_STAMP(25) }
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) gb[vv] = gbn[vv];
    }
    FBSMI_STAMP(30)
    // The log-weights of this workgroup's slots.  Every row tile of a slot was multiplied here, so the slot's row sum (the
    // reference's sum over the observation coordinates, in row order) is taken here too: once the tiles are done with the LDS
    // the lanes park their terms in the G tile's memory as [32 slots][dv + 4] (rows 27 quads apart at d = 100: conflict-free for
    // the 32 readers), and one thread per slot adds its row in order.  The terms never reach global memory and the launch that
    // read them back (k_lgw_lse: 8.5 us at 10 000 particles, 4 MB per chain in uncoalesced 416-byte rows) becomes k_lg_lwpart,
    // the logsumexp partials of the log-weights.  (An earlier form read the rows back from global memory in this tail --
    // stores drained, the CU's L1 dropped: 6 us per workgroup, affordable only in single-round launches.)
    if (VEC4) {
        float* Ls = dyn;
        const int LSs = dvp + 4;
        __syncthreads();   // every wave is done with the tiles
#pragma unroll
        for (int j = 0; j < kVTiles; ++j) {
            const int r0 = kWideTile * (vt0 + j) + 16 * ar + 4 * (lane >> 4);
            if (r0 >= du && r0 < D)
                *reinterpret_cast<float4*>(Ls + jloc * LSs + (r0 - du)) = make_float4(vsv[j][0], vsv[j][1], vsv[j][2], vsv[j][3]);
        }
        __syncthreads();
        if (t < kWideTile) {
            const int m = kWideTile * ts + t;
            if (m < N) {
                const float4* p = reinterpret_cast<const float4*>(Ls + t * LSs);
                float a = 0.0f;
                for (int q = 0; q < (d.dv >> 2); ++q) {   // dv is a multiple of four on this path
                    const float4 x = p[q];
                    a = q == 0 ? x.x : a + x.x;          // (the sum STARTS with the first term: 0 + x would lose the sign of a -0.0)
                    a = a + x.y;
                    a = a + x.z;
                    a = a + x.w;
                }
                d.lw[m] = a;
            }
        }
    }
#ifdef FBSMI_STAMPS
    if (threadIdx.x == 0 && s == d.T - 1) span[1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// (Round 3 built the variant this file's history kept describing -- TWO independent accumulator chains per wave: 64-row G
// tiles, each wave alternating between the accumulators of two row blocks against the same 16 gathered slots, 80 KB of LDS,
// bit-exact -- and measured it on one box against this kernel: d = 100, N = 10 000, two chain groups 22.5 against 19.4 ms per
// sweep, N = 100 000 182 against 174 ms.  With two workgroups per CU every SIMD already holds two waves whose dependent MFMA
// chains interleave, so the matrix pipe was not waiting on the chain; the 64-row tiles only add 14 % padded rows at D = 200.
// Dropped.)

// wide particle filters: initial particles (n, du) row-major -> u0 (same layout) [+ filtering path slot 0]
__global__ void __launch_bounds__(kBlock) k_lgwf_init(LgDev dd, const float* u0s_all) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const float* u0s = u0s_all + (size_t)blockIdx.y * d.N * d.du;
    const size_t tot = (size_t)d.N * d.du;
    for (size_t e = blockIdx.x * (size_t)kBlock + threadIdx.x; e < tot; e += (size_t)gridDim.x * kBlock) {
        const float v = u0s[e];
        d.u0[e] = v;
        if (d.uss) d.uss[e] = v;
    }
}

// wide particle filters with more than 256 particles: stratified / systematic ancestors of every slot from the
// global CDF (resampling.py:43-51; the search of k_filt_prop), resampling key of step kres
__global__ void __launch_bounds__(kBlock) k_lgwf_anc(LgDev dd, int kres) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float heapW[kHeapSize];
    const int N = d.N, m = blockIdx.x * kBlock + threadIdx.x;
    if (threadIdx.x >= 1 && threadIdx.x < kHeapSize) heapW[threadIdx.x] = d.cdf[heap_node_mid(threadIdx.x, N)];
    const uint32_t r0 = d.keytab[8 * kres + 2], r1 = d.keytab[8 * kres + 3];
    __syncthreads();
    if (m < N) {
        const float uu = d.systematic ? uniform_at(r0, r1, 1, 0) : uniform_at(r0, r1, (uint64_t)N, (uint64_t)m);
        const float q = ((float)m + uu) / (float)N;
        int a = bisect_heap(d.cdf, N, d.levels, heapW, q);
        d.anc[m] = a < 0 ? 0 : (a > N - 1 ? N - 1 : a);
    }
}

// ... and the final resampled particles us[inds] (smc.py:72 of the last step) from d.anc
__global__ void __launch_bounds__(kBlock) k_lgwf_gather(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const float* __restrict__ up = (d.T & 1) ? d.u1 : d.u0;
    const size_t tot = (size_t)d.N * d.du;
    for (size_t e = blockIdx.x * (size_t)kBlock + threadIdx.x; e < tot; e += (size_t)gridDim.x * kBlock) {
        const size_t m = e / d.du, r = e - m * d.du;
        const float x = up[(size_t)d.anc[m] * d.du + r];
        d.usT[e] = x;
        if (d.uss) d.uss[(size_t)d.T * tot + e] = x;
    }
}

// bootstrap_filter's last act: normalise the last weights, resample, return us[inds] (smc.py:66-72 of the last step)
__global__ void __launch_bounds__(kBlock) k_lgwf_final(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ LgwPreLds pre;
    lgw_fpre_body<true>(d, d.T - 1, true, pre, []() {});
    const float* __restrict__ up = (d.T & 1) ? d.u1 : d.u0;
    const int tot = d.N * d.du;
    for (int e = threadIdx.x; e < tot; e += kBlock) {
        const int m = e / d.du, r = e - m * d.du;
        const float x = up[(size_t)pre.ancS[m] * d.du + r];
        d.usT[e] = x;
        if (d.uss) d.uss[(size_t)d.T * tot + e] = x;
    }
}

// ------------------------------------------------------------------------------------------
// Narrow models with at most 256 particles: the ensemble is one tile and one workgroup, so a whole SMC
// step -- normalise, both CDFs (in LDS), J, redraws, gather, Euler-Maruyama, pin, log-weight -- is
// ONE launch of one workgroup per chain (the step prologue is shared with the wide path).
// ------------------------------------------------------------------------------------------
template <int DMAX>
__device__ __forceinline__ void lg_step1_body(const LgDev& d, int s, LgwPreLds& pre) {
    const int N = d.N, m = threadIdx.x;
    const bool live = m < N;
    const uint32_t* kt = d.keytab + 8 * s;
    const uint32_t t0 = kt[6], t1 = kt[7];
    const int j_ref = d.bs[s + 1];
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    const StepTables<DMAX> t = step_tables<DMAX>(d, s);
    const float* v_prev = d.vs + (size_t)s * d.dv;
    const float* v = d.vs + (size_t)(s + 1) * d.dv;
    const float* ustar = d.us_star + (size_t)(s + 1) * d.du;
    float xi[DMAX];
    auto draw_noise = [&]() {
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            xi[r] = (r < d.du && live) ? normal_at(t0, t1, (uint64_t)N * d.du, (uint64_t)m * d.du + r) : 0.0f;
    };
    lgw_pre_body<false>(d, s, true, pre, draw_noise);
    if (live) {
        const int a = pre.ancS[m];
        const bool pinned = m == j_ref;
        float u[DMAX];
#pragma unroll
        for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + a] : 0.0f;
        // transition_sampler (gp_gibbs.py:120-122) and the pin of csmc.py:143
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            if (r < d.du) {
                const float dr = drift_row<DMAX>(t, r, u, v_prev);
                float x = (u[r] + dr * t.dt) + t.sd * xi[r];
                if (pinned) x = ustar[r];
                un[(size_t)r * N + m] = x;
                if (d.uss) d.uss[((size_t)(s + 1) * N + m) * d.du + r] = x;
            }
        }
        d.lw[m] = lg_loglik<DMAX>(t, u, v, v_prev);   // likelihood_logpdf on the gathered particle (csmc.py:145)
    }
}

template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_step1(LgDev dd, int s) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ LgwPreLds pre;
    lg_step1_body<DMAX>(d, s, pre);
}

// ... and, since one workgroup owns the chain, all T steps in ONE launch: a workgroup barrier is the only
// synchronisation a step needs (its own stores are visible to its own waves: they share the CU's L1).
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_lg_sweep1(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ LgwPreLds pre;
    for (int s = 0; s < d.T; ++s) {
        lg_step1_body<DMAX>(d, s, pre);
        __threadfence_block();
        __syncthreads();
    }
}

// Narrow particle filters with at most 256 particles: the whole run (bootstrap_filter smc.py:58-86 or
// pmcmc_filter_step smc.py:138-157) in ONE launch of one workgroup per chain -- per step the filter
// prologue (logsumexp, log-likelihood accumulator, weights, cumsum, stratified / systematic ancestors, all in
// LDS) followed by gather, propagate and weight, a workgroup barrier between steps.
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_filt_sweep1(LgDev dd, const float* u0s_all) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ LgwPreLds pre;
    const int N = d.N, m = threadIdx.x, T = d.T;
    const bool live = m < N;
    const float* u0s = u0s_all + (size_t)blockIdx.y * N * d.du;
    // initial particles; pmcmc: log-weights of step 0 on them (smc.py:144, k = 0)
    if (live) {
        float u[DMAX];
#pragma unroll
        for (int r = 0; r < DMAX; ++r) {
            u[r] = r < d.du ? u0s[(size_t)m * d.du + r] : 0.0f;
            if (r < d.du) {
                d.u0[(size_t)r * N + m] = u[r];
                if (d.uss) d.uss[(size_t)m * d.du + r] = u[r];
            }
        }
        if (d.flow == 1) d.lw[m] = lg_loglik<DMAX>(step_tables<DMAX>(d, 0), u, d.vs + d.dv, d.vs);
    }
    __threadfence_block();
    __syncthreads();
    for (int k = 0; k <= T; ++k) {
        const bool last = k == T;             // bootstrap only: the final resampling (smc.py:72 of the last step)
        if (last && d.flow == 1) break;
        const float* __restrict__ up = (k & 1) ? d.u1 : d.u0;
        float* __restrict__ un = (k & 1) ? d.u0 : d.u1;
        const bool resample = d.flow == 1 || k > 0;
        int a = m;
        if (resample) {
            lgw_fpre_body<false>(d, d.flow == 1 ? k : k - 1, true, pre, []() {});
            a = live ? pre.ancS[m] : 0;
        }
        if (live) {
            float u[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + a] : 0.0f;
            if (last) {
#pragma unroll
                for (int r = 0; r < DMAX; ++r)
                    if (r < d.du) {
                        d.usT[(size_t)m * d.du + r] = u[r];
                        if (d.uss) d.uss[((size_t)T * N + m) * d.du + r] = u[r];
                    }
            } else {
                if (d.uss && d.flow == 0 && k > 0) {   // filtering_samples[k] = resampled particles of step k-1
#pragma unroll
                    for (int r = 0; r < DMAX; ++r)
                        if (r < d.du) d.uss[((size_t)k * N + m) * d.du + r] = u[r];
                }
                const StepTables<DMAX> t = step_tables<DMAX>(d, k);
                const uint32_t p0 = d.keytab[8 * k], p1 = d.keytab[8 * k + 1];
                float x[DMAX];
#pragma unroll
                for (int r = 0; r < DMAX; ++r) {
                    x[r] = 0.0f;
                    if (r < d.du) {
                        const float dr = drift_row<DMAX>(t, r, u, d.vs + (size_t)k * d.dv);
                        const float z = normal_at(p0, p1, (uint64_t)N * d.du, (uint64_t)m * d.du + r);
                        x[r] = (u[r] + dr * t.dt) + t.sd * z;                 // transition_sampler
                        un[(size_t)r * N + m] = x[r];
                        if (k == T - 1 && d.flow == 1) d.usT[(size_t)m * d.du + r] = x[r];
                    }
                }
                if (d.flow == 0) {          // measurement_cond_pdf(v, us_prev, v_prev, t_prev)      smc.py:65
                    d.lw[m] = lg_loglik<DMAX>(t, u, d.vs + (size_t)(k + 1) * d.dv, d.vs + (size_t)k * d.dv);
                } else if (k + 1 < T) {     // next step's likelihood_logpdf on the propagated particle  smc.py:144
                    const StepTables<DMAX> tn = step_tables<DMAX>(d, k + 1);
                    d.lw[m] = lg_loglik<DMAX>(tn, x, d.vs + (size_t)(k + 2) * d.dv, d.vs + (size_t)(k + 1) * d.dv);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// the logsumexp tile partials of the stored log-weights (after the last k_lg_step1, for the final-mode kernels)
__global__ void __launch_bounds__(kBlock) k_lg_lwpart(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    const int m = blockIdx.x * kBlock + threadIdx.x;
    float lv[1] = {m < d.N ? d.lw[m] : -__builtin_inff()};
    float mx, sx;
    block_lse_partial<1>(lv, xch[0], xch[1], mx, sx);
    if (threadIdx.x == 0) {
        d.bmax[blockIdx.x] = mx;
        d.bsumexp[blockIdx.x] = sx;
    }
}

// ------------------------------------------------------------------------------------------
// Fused particle filters for the analytic model: bootstrap_filter (fbs/samplers/smc.py:9-88) and
// pmcmc_filter_step (smc.py:115-158) with stratified / systematic resampling
// (fbs/samplers/resampling.py:43-59).  Same three dependency levels per step as the Gibbs sweep:
//   fnorm (lse, w, log-likelihood accumulator, partials of w) -> cdf -> fprop (+ per-workgroup (max, sumexp)).
// The two functions order a step differently:
//   bootstrap : propagate(us_prev) ; weight(us_prev) ; resample the NEW particles
//               => fprop(k) = [resample with step k-1's weights] gather, propagate, weight (old)
//   pmcmc     : weight(us) ; resample ; propagate
//               => fprop(k) = resample, gather, propagate, weight the NEW particle for step k+1
// ------------------------------------------------------------------------------------------
// keys: bootstrap: key -> (key_init, key_steps); keys = split(key_steps, T)        smc.py:77-79
//       pmcmc    : keys = split(key, T)                                            smc.py:154
//       step     : keys[k] -> (key_proposal, key_resampling)                       smc.py:61,142
__global__ void __launch_bounds__(kBlock) k_filt_keys(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    uint32_t s0 = d.keys[0], s1 = d.keys[1];
    if (d.flow == 0) split_at(d.keys[0], d.keys[1], 2, 1, s0, s1);
    for (int k = threadIdx.x; k < d.T; k += kBlock) {
        uint32_t q0, q1;
        split_at(s0, s1, d.T, k, q0, q1);
        uint32_t* kt = d.keytab + 8 * k;
        split_at(q0, q1, 2, 0, kt[0], kt[1]);  // key_proposal
        split_at(q0, q1, 2, 1, kt[2], kt[3]);  // key_resampling
    }
    if (threadIdx.x == 0) *d.ell = 0.0f;
}

// u0s (n, du) row-major -> u0 (SoA); pmcmc: log-weights of step 0 on the initial particles
template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_filt_init(LgDev dd, const float* u0s_all) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    const float* u0s = u0s_all + (size_t)blockIdx.y * d.N * d.du;
    const StepTables<DMAX> t = step_tables<DMAX>(d, 0);
    const int base = (blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float lv[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = base + i;
        lv[i] = -__builtin_inff();
        if (p < d.N) {
            float u[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                u[r] = r < d.du ? u0s[(size_t)p * d.du + r] : 0.0f;
                if (r < d.du) {
                    d.u0[(size_t)r * d.N + p] = u[r];
                    if (d.uss) d.uss[(size_t)p * d.du + r] = u[r];
                }
            }
            if (d.flow == 1) {
                const float l = lg_loglik<DMAX>(t, u, d.vs + d.dv, d.vs);   // smc.py:144, k = 0
                d.lw[p] = l;
                lv[i] = l;
            }
        }
    }
    if (d.flow == 1) {
        float mx, sx;
        block_lse_partial<ITEMS>(lv, xch[0], xch[1], mx, sx);
        if (threadIdx.x == 0) {
            d.bmax[blockIdx.x] = mx;
            d.bsumexp[blockIdx.x] = sx;
        }
    }
}

// fnorm: c = logsumexp(lw); accumulate the (negative) log-likelihood; w = exp(lw - c); partials
template <int ITEMS, bool PUB = false>   // PUB: publish the tile's part of the summation tree (two-launch filter step)
__global__ void __launch_bounds__(kBlock) k_filt_norm(LgDev dd) {
    int bx = blockIdx.x;
    if (PUB && dd.pin) {   // small ensembles: every eighth block works (one XCD), see LgDev.pin
        if (blockIdx.x & 7) return;
        bx = blockIdx.x >> 3;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[3][4];
    const int base = (bx * kBlock + threadIdx.x) * ITEMS;
    float l[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) l[i] = base + i < d.N ? d.lw[base + i] : 0.0f;
    float c, Mraw;
    lse_from_partials(d.bmax, d.bsumexp, d.nb, xch[0], xch[1], c, Mraw);
    float xw[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int e = base + i;
        xw[i] = 0.0f;
        if (e < d.N) {
            const float w = fbsmi_expf(l[i] - c);   // smc.py:68-69 / :147-148
            d.w[e] = w;
            xw[i] = w;
        }
    }
    float s1[1] = {chunk_total<ITEMS>(xw)}, t1[1];
    TreePath p1[1];
    block_upsweep_n<1>(s1, p1, xch[2], t1);
    if (PUB && ITEMS == 1) {   // as k_lg_norm<1, 0, true>: the midpoints of the nodes of 8 leaves and more
        const int i = threadIdx.x;
        if ((i & 3) == 0) {
            const int h = i ? tree_mid_node(i) : 0;
            const float2 node = make_float2(i ? tree_left_sum(p1[0], i) : 0.0f, xw[0]);
            d.trW[(size_t)bx * kTreeNodes + h] = node;
            if (h < kMidN) d.trWtop[bx * kMidN + h] = node;
            if (i == 0) d.wfirst[bx] = node.y;
        }
    }
    if (threadIdx.x == 0) {
        d.bsumw[bx] = t1[0];
        if (bx == 0) {
            const float e0 = *d.ell;
            *d.ell = d.flow == 0 ? e0 - (c - d.logn)      // log_nell -= _c - log(N)          smc.py:67
                                 : (e0 - d.logn) + c;      // log_ell = log_ell - log(N) + _c  smc.py:146
        }
    }
}

// fprop.  RESAMPLE: draw ancestors from the current cdf with resample key of step `kres`.
// WEIGHT: 0 none, 1 weight the gathered (old) particle with step s tables, 2 weight the new
// particle with step s+1 tables.  PROPAGATE: false = gather only (the final resampling of
// bootstrap_filter).
template <int ITEMS, int DMAX>
__global__ void __launch_bounds__(kBlock) k_filt_prop(LgDev dd, int s, int resample, int kres, int weight,
                                                      int propagate) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[2][4];
    __shared__ float heapW[kHeapSize];
    const int N = d.N;
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    float last = 0.0f;
    if (resample) {
        last = d.cdf[N - 1];
        if (threadIdx.x >= 1 && threadIdx.x < kHeapSize) heapW[threadIdx.x] = d.cdf[heap_node_mid(threadIdx.x, N)];
    }
    (void)last;
    const uint32_t* ktp = d.keytab + 8 * (s < d.T ? s : d.T - 1);
    const uint32_t p0 = ktp[0], p1 = ktp[1];
    const uint32_t r0 = d.keytab[8 * kres + 2], r1 = d.keytab[8 * kres + 3];
    const float u_sys = (resample && d.systematic) ? uniform_at(r0, r1, 1, 0) : 0.0f;
    __syncthreads();
    const int st = s < d.T ? s : d.T - 1;
    const StepTables<DMAX> t = step_tables<DMAX>(d, st);
    const StepTables<DMAX> tn = step_tables<DMAX>(d, st + 1 < d.T ? st + 1 : st);
    const int base = (blockIdx.x * kBlock + threadIdx.x) * ITEMS;
    float lv[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int m = base + i;
        lv[i] = -__builtin_inff();
        if (m < N) {
            int a = m;
            if (resample) {   // _systematic_or_stratified, resampling.py:43-51
                const float uu = d.systematic ? u_sys : uniform_at(r0, r1, (uint64_t)N, (uint64_t)m);
                const float q = ((float)m + uu) / (float)N;
                a = bisect_heap(d.cdf, N, d.levels, heapW, q);
                a = a < 0 ? 0 : (a > N - 1 ? N - 1 : a);
            }
            float u[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + a] : 0.0f;
            if (!propagate) {   // final resampling of bootstrap_filter: us[inds]  (smc.py:72)
#pragma unroll
                for (int r = 0; r < DMAX; ++r)
                    if (r < d.du) {
                        d.usT[(size_t)m * d.du + r] = u[r];
                        if (d.uss) d.uss[((size_t)d.T * N + m) * d.du + r] = u[r];
                    }
                continue;
            }
            if (d.uss && d.flow == 0 && s > 0) {   // filtering_samples[s] = resampled particles of step s-1
#pragma unroll
                for (int r = 0; r < DMAX; ++r)
                    if (r < d.du) d.uss[((size_t)s * N + m) * d.du + r] = u[r];
            }
            float x[DMAX];
#pragma unroll
            for (int r = 0; r < DMAX; ++r) {
                x[r] = 0.0f;
                if (r < d.du) {
                    const float dr = drift_row<DMAX>(t, r, u, d.vs + (size_t)s * d.dv);
                    const float z = normal_at(p0, p1, (uint64_t)N * d.du, (uint64_t)m * d.du + r);
                    x[r] = (u[r] + dr * t.dt) + t.sd * z;                 // transition_sampler
                    un[(size_t)r * N + m] = x[r];
                    if (s == d.T - 1 && d.flow == 1) d.usT[(size_t)m * d.du + r] = x[r];
                }
            }
            if (weight == 1) {         // measurement_cond_pdf(v, us_prev, v_prev, t_prev)      smc.py:65
                const float l = lg_loglik<DMAX>(t, u, d.vs + (size_t)(s + 1) * d.dv, d.vs + (size_t)s * d.dv);
                d.lw[m] = l;
                lv[i] = l;
            } else if (weight == 2) {  // next step's likelihood_logpdf on the propagated particle  smc.py:144
                const float l = lg_loglik<DMAX>(tn, x, d.vs + (size_t)(s + 2) * d.dv, d.vs + (size_t)(s + 1) * d.dv);
                d.lw[m] = l;
                lv[i] = l;
            }
        }
    }
    if (weight) {
        float mx, sx;
        block_lse_partial<ITEMS>(lv, xch[0], xch[1], mx, sx);
        if (threadIdx.x == 0) {
            d.bmax[blockIdx.x] = mx;
            d.bsumexp[blockIdx.x] = sx;
        }
    }
}

// fprop of the TWO-launch filter step (N a power of two, 2..256 tiles; k_filt_norm<1, true> in front, no cdf launch): the
// stratified / systematic searches walk the summation tree, exactly as the Cat(w) redraws of k_lg_prop1t do -- the top
// levels rebuilt from the tile sums, three levels of every tile staged in LDS, then the tile's published heap and the last
// four leaves of w in two round trips.  Same flags and arithmetic as k_filt_prop<1, DMAX>.
template <int DMAX>
__global__ void __launch_bounds__(kBlock) k_filt_prop1t(LgDev dd, int s, int resample, int kres, int weight, int propagate) {
    int bx = blockIdx.x;
    if (dd.pin) {
        if (blockIdx.x & 7) return;
        bx = blockIdx.x >> 3;
    }
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ float xch[3][4];
    __shared__ __attribute__((aligned(16))) float2 topW[kBlock];
    __shared__ __attribute__((aligned(16))) float2 midW[kMidN * kBlock];
    const int N = d.N, nb = d.nb, tid = threadIdx.x;
    const float* __restrict__ up = (s & 1) ? d.u1 : d.u0;
    float* __restrict__ un = (s & 1) ? d.u0 : d.u1;
    float last = 0.0f;
    if (resample) {   // (kernel-uniform)
        const bool tl = tid < nb;
        const float sw = tl ? d.bsumw[tid] : 0.0f, wf = tl ? d.wfirst[tid] : 0.0f;
#pragma unroll
        for (int k = 0; k < kMidN / 2; ++k) {
            const int idx = tid + kBlock * k;
            if (idx < kMidN / 2 * nb) reinterpret_cast<float4*>(midW)[idx] = reinterpret_cast<const float4*>(d.trWtop)[idx];
        }
        float s1[1] = {sw}, t1[1];
        TreePath p1[1];
        block_upsweep_n<1>(s1, p1, xch[2], t1);
        last = t1[0];                                  // == cdf[N - 1]
        if (tid) topW[tree_mid_node(tid)] = make_float2(tree_left_sum(p1[0], tid), wf);
        __syncthreads();
    }
    const uint32_t* ktp = d.keytab + 8 * (s < d.T ? s : d.T - 1);
    const uint32_t p0 = ktp[0], p1k = ktp[1];
    const uint32_t r0 = d.keytab[8 * kres + 2], r1 = d.keytab[8 * kres + 3];
    const float u_sys = (resample && d.systematic) ? uniform_at(r0, r1, 1, 0) : 0.0f;
    const int st = s < d.T ? s : d.T - 1;
    const StepTables<DMAX> t = step_tables<DMAX>(d, st);
    const StepTables<DMAX> tn = step_tables<DMAX>(d, st + 1 < d.T ? st + 1 : st);
    const int m = bx * kBlock + tid;   // N is a multiple of the tile: every slot is live
    float lv[1] = {-__builtin_inff()};
    int a = m;
    if (resample) {   // _systematic_or_stratified, resampling.py:43-51
        const float uu = d.systematic ? u_sys : uniform_at(r0, r1, (uint64_t)N, (uint64_t)m);
        const float q = ((float)m + uu) / (float)N;
        float P = 0.0f, E = last;
        int h = tree_walk_n(topW, kBlock / nb, 31 - __builtin_clz(nb), q, P, E);
        const int tile = h - kBlock;
        h = tree_walk_n(midW + tile * kMidN, 1, kMidLv, q, P, E);
        const TreeRound rd = tree_round_load(d, tile, h);
        const int lo = tree_round_walk(rd, tile, h, q, P, E);
        const float4 w4 = *reinterpret_cast<const float4*>(d.w + lo);
        a = tree_leaves_walk(w4, lo, q, P, E);
        a = a < 0 ? 0 : (a > N - 1 ? N - 1 : a);
    }
    float u[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) u[r] = r < d.du ? up[(size_t)r * N + a] : 0.0f;
    if (!propagate) {   // final resampling of bootstrap_filter: us[inds]  (smc.py:72)
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            if (r < d.du) {
                d.usT[(size_t)m * d.du + r] = u[r];
                if (d.uss) d.uss[((size_t)d.T * N + m) * d.du + r] = u[r];
            }
        return;   // (kernel-uniform; weight == 0 here)
    }
    if (d.uss && d.flow == 0 && s > 0) {   // filtering_samples[s] = resampled particles of step s-1
#pragma unroll
        for (int r = 0; r < DMAX; ++r)
            if (r < d.du) d.uss[((size_t)s * N + m) * d.du + r] = u[r];
    }
    float x[DMAX];
#pragma unroll
    for (int r = 0; r < DMAX; ++r) {
        x[r] = 0.0f;
        if (r < d.du) {
            const float dr = drift_row<DMAX>(t, r, u, d.vs + (size_t)s * d.dv);
            const float z = normal_at(p0, p1k, (uint64_t)N * d.du, (uint64_t)m * d.du + r);
            x[r] = (u[r] + dr * t.dt) + t.sd * z;                 // transition_sampler
            un[(size_t)r * N + m] = x[r];
            if (s == d.T - 1 && d.flow == 1) d.usT[(size_t)m * d.du + r] = x[r];
        }
    }
    if (weight == 1) {         // measurement_cond_pdf(v, us_prev, v_prev, t_prev)      smc.py:65
        const float l = lg_loglik<DMAX>(t, u, d.vs + (size_t)(s + 1) * d.dv, d.vs + (size_t)s * d.dv);
        d.lw[m] = l;
        lv[0] = l;
    } else if (weight == 2) {  // next step's likelihood_logpdf on the propagated particle  smc.py:144
        const float l = lg_loglik<DMAX>(tn, x, d.vs + (size_t)(s + 2) * d.dv, d.vs + (size_t)(s + 1) * d.dv);
        d.lw[m] = l;
        lv[0] = l;
    }
    if (weight) {
        float mx, sx;
        block_lse_partial<1>(lv, xch[0], xch[1], mx, sx);
        if (threadIdx.x == 0) {
            d.bmax[bx] = mx;
            d.bsumexp[bx] = sx;
        }
    }
}

// ------------------------------------------------------------------------------------------
// sweep epilogue
// ------------------------------------------------------------------------------------------
// explicit backward (gibbs.py:152-154): force_move tail, x0 = uss[-1, idx]
__global__ void k_lg_force_move(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const float* uT = (d.T & 1) ? d.u1 : d.u0;
    __shared__ int s_idx;
    if (threadIdx.x == 0) {
        const int N = d.N;
        uint32_t p0, p1, q0, q1;
        split_at(d.misc[4], d.misc[5], 2, 0, p0, p1);
        split_at(d.misc[4], d.misc[5], 2, 1, q0, q1);
        const int k = d.bs[d.T];
        const float u1 = uniform_at(p0, p1, 1, 0);
        const int i = searchsorted_left(d.cdf, N, d.levels, d.cdf[N - 1] * (1.0f - u1));
        const float u = uniform_at(q0, q1, 1, 0);
        const float temp = 1.0f - d.w[k];
        const bool accept = u * (1.0f - d.w[i]) < temp;
        s_idx = accept ? i : k;
    }
    __syncthreads();
    const int idx = s_idx;
    for (int r = threadIdx.x; r < d.du; r += blockDim.x) d.x0n[r] = uT[u_at(d, r, idx)];
}

// backward scanning (csmc.py:230-270): B_T ~ Cat(w_T), B_{k-1} = A_k[B_k], x_k = uss[k, B_k]
__global__ void k_lg_backscan(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    __shared__ int s_B[1];
    const int N = d.N;
    if (threadIdx.x == 0) {
        const float u = uniform_at(d.misc[10], d.misc[11], 1, 0);
        int B = searchsorted_left(d.cdf, N, d.levels, d.cdf[N - 1] * (1.0f - u));
        d.bsn[d.T] = B;
        for (int k = d.T; k >= 1; --k) {
            B = d.As[(size_t)(k - 1) * N + B];
            d.bsn[k - 1] = B;
        }
        s_B[0] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (d.T + 1) * d.du; e += blockDim.x) {
        const int k = e / d.du, r = e - k * d.du;
        const float x = d.uss[((size_t)k * N + d.bsn[k]) * d.du + r];
        d.usn[e] = x;
        if (k == d.T) d.x0n[r] = x;
    }
    for (int k = threadIdx.x; k <= d.T; k += blockDim.x) d.acc[k] = d.bsn[k] != d.bs[k];
}

// row-major copy of the final particles (parity view)
__global__ void k_lg_export(LgDev dd) {
    const LgDev d = chain_view(dd, blockIdx.y);
    const float* uT = (d.T & 1) ? d.u1 : d.u0;
    const size_t tot = (size_t)d.N * d.du;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / d.du, r = e - p * d.du;
        d.usT[e] = uT[u_at(d, (int)r, (int)p)];
    }
}

// chain step, one workgroup per chain: x0 <- x0_next, bs <- bs_next, x0s[counter][c] = x0_next;
// chain 0 also advances the master key (key = split(key)[0]) and the sweep counter.
__global__ void k_lg_advance(LgDev dd) {
    const int c = blockIdx.y;
    const LgDev d = chain_view(dd, c);
    const int cnt = *d.counter;
    float* x0s = d.x0s_slot ? *d.x0s_slot : nullptr;
    for (int r = threadIdx.x; r < d.du; r += blockDim.x) {
        const float x = d.x0n[r];
        d.x0[r] = x;
        if (x0s) x0s[((size_t)cnt * d.Ctot + d.c0 + c) * d.du + r] = x;
    }
    for (int k = threadIdx.x; k <= d.T; k += blockDim.x) d.bs[k] = d.bsn[k];
}

__global__ void k_lg_advance_key(LgDev d) {
    if (threadIdx.x == 0) {
        uint32_t a0, a1;
        split_at(d.key[0], d.key[1], 2, 0, a0, a1);
        d.key[0] = a0;
        d.key[1] = a1;
        *d.counter = *d.counter + 1;
    }
}

}  // namespace fbsmi

using namespace fbsmi;

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// The pinned launches (LgDev.pin, FBSMI_WIDE_PIN) rest on an OBSERVED dispatch order -- workgroups are dealt round-robin over
// the eight XCDs, so blocks b and b + 8 share one -- that HIP does not promise.  It is checked once per process on the
// hardware itself: a 64-block probe launch reads HW_REG_XCC_ID in every block; pinning stays on only if blocks of equal
// b % 8 all report the same XCD and the eight classes report eight different ones.  (Results never depend on it: a pinned
// launch only chooses WHICH blocks do the work.)  FBSMI_PIN_CHECK=0 skips the probe and trusts the order.
__global__ void k_xcc_probe(int* out) {
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = (int)(id & 15);
    }
}

static bool xcd_round_robin_holds() {
    static const int ok = [] {
        if (const char* e = getenv("FBSMI_PIN_CHECK"))
            if (atoi(e) == 0) return 1;
        constexpr int kProbeBlocks = 64;
        int* dev = nullptr;
        int host[kProbeBlocks];
        if (hipMalloc(&dev, sizeof(host)) != hipSuccess) return 0;
        bool good = hipMemset(dev, 0xff, sizeof(host)) == hipSuccess;
        if (good) {
            k_xcc_probe<<<kProbeBlocks, 64, 0, nullptr>>>(dev);
            good = hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
        }
        (void)hipFree(dev);
        if (!good) return 0;
        unsigned seen = 0;
        for (int b = 0; b < kProbeBlocks; ++b) {
            if (host[b] < 0 || host[b] > 15 || host[b] != host[b & 7]) return 0;
            if (b < 8) seen |= 1u << host[b];
        }
        return __builtin_popcount(seen) == 8 ? 1 : 0;
    }();
    return ok != 0;
}

// Launch streams come from a process-wide pool of four, created once and never destroyed: handle (group) g of a batch uses
// stream g % 4.  One stream per handle made the number of streams -- and with it the hardware queue a stream lands on -- depend
// on how many handles a process had created before: with the queues oversubscribed two chain groups could end up sharing one
// and a 16 us step took 29 (round 3: the d = 100 toy at 100 particles measured 3.0 ms per sweep alone and 5.9 ms when a
// smaller ensemble's handles had been created first).
constexpr int kStreamPool = 4;
static hipStream_t pool_stream(int g) {
    static hipStream_t pool[kStreamPool] = {nullptr, nullptr, nullptr, nullptr};
    static int dev_of[kStreamPool] = {-1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int i = ((g % kStreamPool) + kStreamPool) % kStreamPool;
    if (!pool[i] || dev_of[i] != dev) {   // (one device per process is the rule: one rank per GPU; a second device gets fresh streams)
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
        pool[i] = st;
        dev_of[i] = dev;
    }
    return pool[i];
}

struct fbsmi_lg_sweep {
    LgDev d{};
    int items = 1, dmax = 2;
    std::vector<void*> allocs;
    std::vector<std::pair<void**, size_t>> slab_reqs;   // (pointer slot, offset) until slab_commit
    size_t slab_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraphExec_t graph_single = nullptr;  // one sweep, no chain bookkeeping
    hipGraphExec_t graph_chain = nullptr;   // one sweep + key split + advance
    bool profile = false;
    int two_slot_prop = -1;  // FBSMI_TWO_SLOT_PROP=0|1: never / always (where applicable) k_lg_prop2; unset: by batch size
    bool step_launches = false;  // FBSMI_STEP_LAUNCHES=1: one launch per step also where one launch per sweep is possible
    bool generic_prop = false;  // FBSMI_GENERIC_PROP=1: k_lg_prop also for one slot per thread (timing experiments)
    bool tree_step = true;  // FBSMI_TREE_STEP=0: keep the cdf launch also where the two-launch step applies
    int tree_halves = -1;   // FBSMI_TREE_HALVES=1|2|4: tiles per workgroup of k_lg_prop1t (unset: as many as there are tiles per CU, up to 4)
    int debug_mask = 7;  // FBSMI_DEBUG_STEP_MASK: bit0 norm, bit1 cdf, bit2 prop (timing experiments only)
    std::vector<hipEvent_t> prof_ev[kNumProfKernels];  // pairs (start, stop)
    double prof_us[kNumProfKernels] = {0, 0, 0};
    int64_t prof_n[kNumProfKernels] = {0, 0, 0};
};

namespace {

template <typename T>
int dev_alloc(fbsmi_lg_sweep* s, T** p, size_t count) {
    void* q = nullptr;
    FBSMI_HIP_TRY(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    FBSMI_HIP_TRY(hipMemset(q, 0, sizeof(T) * (count ? count : 1)));
    s->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

// The buffers of a handle are carved from ONE allocation: the step kernels touch ~20 of them per
// launch, and as separate hipMallocs each sits in pages of its own -- a first-touch translation miss
// per buffer per kernel, on the critical path of kernels that run for a few microseconds.
template <typename T>
int slab_request(fbsmi_lg_sweep* s, T** p, size_t count) {
    const size_t bytes = (sizeof(T) * (count ? count : 1) + 255) & ~(size_t)255;
    s->slab_reqs.push_back({(void**)p, s->slab_bytes});
    s->slab_bytes += bytes;
    *p = nullptr;
    return 0;
}

int slab_commit(fbsmi_lg_sweep* s) {
    void* q = nullptr;
    FBSMI_HIP_TRY(hipMalloc(&q, s->slab_bytes ? s->slab_bytes : 256));
    s->allocs.push_back(q);
    FBSMI_HIP_TRY(hipMemset(q, 0, s->slab_bytes ? s->slab_bytes : 256));
    for (auto& r : s->slab_reqs) *r.first = (char*)q + r.second;
    s->slab_reqs.clear();
    return 0;
}

#define LG_DISPATCH(s, ...)                                                                   \
    do {                                                                                      \
        if ((s)->items == 1) {                                                                \
            constexpr int ITEMS = 1;                                                          \
            if ((s)->dmax == 1) { constexpr int DMAX = 1; __VA_ARGS__; }                      \
            else if ((s)->dmax == 2) { constexpr int DMAX = 2; __VA_ARGS__; }                 \
            else if ((s)->dmax == 4) { constexpr int DMAX = 4; __VA_ARGS__; }                 \
            else { constexpr int DMAX = 16; __VA_ARGS__; }                                    \
        } else if ((s)->items == 4) {                                                         \
            constexpr int ITEMS = 4;                                                          \
            if ((s)->dmax == 1) { constexpr int DMAX = 1; __VA_ARGS__; }                      \
            else if ((s)->dmax == 2) { constexpr int DMAX = 2; __VA_ARGS__; }                 \
            else if ((s)->dmax == 4) { constexpr int DMAX = 4; __VA_ARGS__; }                 \
            else { constexpr int DMAX = 16; __VA_ARGS__; }                                    \
        } else {                                                                              \
            constexpr int ITEMS = 16;                                                         \
            if ((s)->dmax == 1) { constexpr int DMAX = 1; __VA_ARGS__; }                      \
            else if ((s)->dmax == 2) { constexpr int DMAX = 2; __VA_ARGS__; }                 \
            else if ((s)->dmax == 4) { constexpr int DMAX = 4; __VA_ARGS__; }                 \
            else { constexpr int DMAX = 16; __VA_ARGS__; }                                    \
        }                                                                                     \
    } while (0)

struct ProfScope {
    fbsmi_lg_sweep* s;
    int which;
    hipStream_t st;
    ProfScope(fbsmi_lg_sweep* s_, int which_, hipStream_t st_) : s(s_), which(which_), st(st_) {
        if (s->profile) {
            hipEvent_t e;
            hipEventCreate(&e);
            hipEventRecord(e, st);
            s->prof_ev[which].push_back(e);
        }
    }
    ~ProfScope() {
        if (s->profile) {
            hipEvent_t e;
            hipEventCreate(&e);
            hipEventRecord(e, st);
            s->prof_ev[which].push_back(e);
        }
    }
};

// the launch sequence of one sweep on stream st
int enqueue_sweep(fbsmi_lg_sweep* s, hipStream_t st, int chain) {
    const LgDev& d = s->d;
    const int nb = d.nb;
    const dim3 gone(1, d.C), gtile(nb, d.C);
    k_lg_keys<<<gone, kBlock, 0, st>>>(d, chain);
    {
        const int64_t n = (int64_t)d.T * d.D;
        int g = (int)((n + 255) / 256);
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        k_lg_noise<<<dim3(g, d.C), 256, 0, st>>>(d);
    }
    const int gpath = (d.D + 63) / 64;
    k_lg_path<<<dim3(gpath, d.C), 64, 0, st>>>(d, 0);
    // wide models: MFMA drift, one workgroup per (32 slots, 32 drift rows)
    const int w_nrt = (d.D + kWideTile - 1) / kWideTile, w_Kp = (d.D + 15) / 16 * 16;
    const int w_S = wide_plane_row(w_Kp);
    const size_t w_lds = wide_lds_bytes(w_S);
    const dim3 gwide(((d.N + kWideTile - 1) / kWideTile) * w_nrt, d.C);
    if (d.wide) {
        k_lgw_init<<<gtile, kBlock, 0, st>>>(d);
        if (d.ef) {   // initial log-weights: rows >= du of the drift product on the initial particles, no resampling
            const int v_tile0 = d.du / kWideTile;
            k_lgw_gemm<3><<<dim3(((d.N + kWideTile - 1) / kWideTile) * (w_nrt - v_tile0), d.C), kBlock, w_lds, st>>>(
                d, 0, v_tile0, w_nrt - v_tile0, w_Kp, w_S, 2, 1);
            if (d.N > kBlock) k_lgw_lse<<<gtile, kBlock, 0, st>>>(d);   // several tiles: lw + partials for k_lg_norm
        }
    } else LG_DISPATCH(s, (k_lg_init<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d)));
    const bool one_tile = d.N <= kBlock && !s->generic_prop;   // the steps need no grid-wide stage of their own
    const bool persistent = one_tile && !d.wide && !s->profile && !s->step_launches;
    if (persistent) LG_DISPATCH(s, (void)ITEMS; (k_lg_sweep1<DMAX><<<gone, kBlock, 0, st>>>(d)));
    for (int k = 0; one_tile && !persistent && k < d.T; ++k) {
        ProfScope p(s, 2, st);
        if (d.wide) {
            static const int pin = [] { const char* e = getenv("FBSMI_WIDE_PIN"); return (e ? atoi(e) : 1) && xcd_round_robin_holds(); }();
            // (one chain per launch only: with two chains per launch and two chain groups in flight the pinned step measured
            // 29 us against 16 unpinned, round 3)
            static const int pin_multi = [] { const char* e = getenv("FBSMI_WIDE_PIN_MULTI"); return e ? atoi(e) : 0; }();
            if (pin && (d.C == 1 || (pin_multi && d.C <= 4)) && gwide.x <= 32)
                k_lgw_gemm<1><<<dim3(gwide.x * 8, d.C), kBlock, w_lds, st>>>(d, k, 0, w_nrt, w_Kp, w_S, 3, 8 + ((2 * d.c0) & 7));
            else {
                // (+ extra blocks that draw the next step's noise beside the step: one per 1024 elements, at most 16)
                const int64_t tot = (int64_t)d.N * d.du;
                static const int extra_on = [] { const char* e = getenv("FBSMI_WIDE_EXTRA"); return e ? atoi(e) : 1; }();
                const int extra = !extra_on ? 0 : (int)((tot + 1023) / 1024 < 16 ? (tot + 1023) / 1024 : 16);
                k_lgw_gemm<1><<<dim3(gwide.x + extra, d.C), kBlock, w_lds, st>>>(d, k, 0, w_nrt, w_Kp, w_S, 3, 0);
            }
        }
        else LG_DISPATCH(s, (void)ITEMS; (k_lg_step1<DMAX><<<gone, kBlock, 0, st>>>(d, k)));
    }
    if (one_tile) {   // log-weights / tile partial for the final-mode kernels
        if (d.wide) k_lgw_lse<<<gtile, kBlock, 0, st>>>(d);
        else k_lg_lwpart<<<gtile, kBlock, 0, st>>>(d);
    }
    // enough workgroups that instruction issue, not latency, bounds the step (measured crossover: between four and six
    // 256-slot workgroups per CU): two slots per thread, three Threefry calls instead of six
    const bool two_slot = !d.wide && s->items == 1 && !s->generic_prop && d.N % (2 * kBlock) == 0 &&
                          (s->two_slot_prop == 1 || (s->two_slot_prop < 0 && (int64_t)nb * d.C >= 5 * 256));
    // N a power of two: the searches walk the summation tree, no cdf launch (k_lg_prop1t)
    const bool tree = s->tree_step && d.trW && !s->generic_prop;
    // Large wide ensembles (the fat drift kernel): the step's noise is drawn by extra blocks of the three small launches in front
    // of the drift kernel -- norm, cdf and the ancestor search need a few dozen workgroups each and leave the chip empty --
    // instead of a launch of its own (k_lgw_noise: ~10 us per step at 10 000 particles).  FBSMI_WIDE_NOISE_FOLD=0: the own launch.
    // The fat kernel (one workgroup per slot tile walks all row tiles: the gather once per slot tile, noise by the small
    // launches, row sums in its tail, no k_lgw_lse) from ~700 tiled workgroups per launch: d = 100, 4 chains in two groups:
    // 2000 particles 8.6 against 9.3 ms per sweep, 4000: 9.2 against 11.4; 1000: 8.4 against 7.8 (FBSMI_FAT_MIN moves it).
    static const int fat_min = [] { const char* e = getenv("FBSMI_FAT_MIN"); return e ? atoi(e) : 700; }();
    const bool fat = d.wide && (int64_t)gwide.x * d.C > fat_min;
    static const int noise_fold = [] { const char* e = getenv("FBSMI_WIDE_NOISE_FOLD"); return e ? atoi(e) : 1; }();
    const bool fold = fat && noise_fold && s->items == 1 && (s->debug_mask & 7) == 7;
    LgDev dz = d;
    dim3 gz = gtile;
    if (fold) {
        const int64_t third = (((int64_t)d.N * d.du + 1) / 2 + 2) / 3;
        const int64_t want = (third + kBlock - 1) / kBlock;
        dz.nz = (int)(want < 1024 ? want : 1024);
        gz = dim3(gtile.x + dz.nz, gtile.y);
    }
    for (int k = 0; !one_tile && k < d.T; ++k) {
        if (s->debug_mask & 1) {
            ProfScope p(s, 0, st);
            if (tree) k_lg_norm<1, 0, true><<<dim3(gtile.x * (d.pin ? 8 : 1), d.C), kBlock, 0, st>>>(d, k);
            else LG_DISPATCH(s, (void)DMAX; (k_lg_norm<ITEMS, 0><<<gz, kBlock, 0, st>>>(dz, k)));
        }
        if ((s->debug_mask & 2) && !tree) {
            ProfScope p(s, 1, st);
            LG_DISPATCH(s, (void)DMAX; (k_lg_cdf<ITEMS, 0><<<gz, kBlock, 0, st>>>(dz, k)));
        }
        if (s->debug_mask & 4) {
            ProfScope p(s, 2, st);
            if (d.wide) {
                k_lgw_anc<<<gz, kBlock, 0, st>>>(dz, k);
                bool fat_rowsum = false;
                if (fat) {
                    const int64_t pairs = ((int64_t)d.N * d.du + 1) / 2;
                    if (!fold)
                        k_lgw_noise<<<dim3((unsigned)((pairs + kBlock - 1) / kBlock < 4096 ? (pairs + kBlock - 1) / kBlock : 4096), d.C),
                                      kBlock, 0, st>>>(d, k);
                    const int nst = (d.N + kWideTile - 1) / kWideTile;
                    fat_rowsum = (d.D & 3) == 0 && (d.du & 3) == 0;   // the float4 kernel leaves the log-weights in d.lw
                    if (fat_rowsum)
                        k_lgw_gemm_fat<true><<<dim3(nst, d.C), kBlock, w_lds, st>>>(d, k, w_nrt, w_Kp, w_S);
                    else
                        k_lgw_gemm_fat<false><<<dim3(nst, d.C), kBlock, w_lds, st>>>(d, k, w_nrt, w_Kp, w_S);
                }
                else
                    k_lgw_gemm<0><<<gwide, kBlock, w_lds, st>>>(d, k, 0, w_nrt, w_Kp, w_S, 3, 0);
                if (fat_rowsum) k_lg_lwpart<<<gtile, kBlock, 0, st>>>(d);   // tile partials of the log-weights the drift kernel left
                else k_lgw_lse<<<gtile, kBlock, 0, st>>>(d);
            } else if (tree && two_slot && !d.plus1) {
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop2t<DMAX><<<dim3(nb / 2, d.C), kBlock, 0, st>>>(d, k)));
            } else if (tree && !d.plus1 && nb % 4 == 0 && (s->tree_halves == 4 || (s->tree_halves < 0 && (int64_t)nb * d.C >= 4 * 256))) {
                // four tiles per 1024-thread workgroup once there are four tiles per CU (measured 8.57 against 8.80 ms per
                // 4-chain sweep with two; a single chain is fastest with one tile per workgroup)
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop1t<DMAX, 4><<<dim3(nb / 4, d.C), 4 * kBlock, 0, st>>>(d, k)));
            } else if (tree && !d.plus1 && nb % 2 == 0 && (s->tree_halves == 2 || (s->tree_halves < 0 && (int64_t)nb * d.C >= 2 * 256))) {
                // two adjacent tiles per 512-thread workgroup: half the waves skip the tree building (measured +2 % at 4 chains,
                // -1 % for a single chain, which keeps one tile per workgroup)
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop1t<DMAX, 2><<<dim3(nb / 2, d.C), 2 * kBlock, 0, st>>>(d, k)));
            } else if (tree) {
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop1t<DMAX, 1><<<dim3(gtile.x * (d.pin ? 8 : 1), d.C), kBlock, 0, st>>>(d, k)));
            } else if (two_slot) {
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop2<DMAX><<<dim3(nb / 2, d.C), kBlock, 0, st>>>(d, k)));
            } else if (s->items == 1 && !s->generic_prop) {
                LG_DISPATCH(s, (void)ITEMS; (k_lg_prop1<DMAX><<<gtile, kBlock, 0, st>>>(d, k)));
            } else if (s->items > 1 && !s->generic_prop) {
                // several slots per thread: compact heaps by their own small launch, then the batched kernel
                k_lg_heaps<<<dim3(kHeapSizeW / kBlock, d.C), kBlock, 0, st>>>(d);
                static const int queue = [] { const char* e = getenv("FBSMI_BIGN_QUEUE"); return e ? atoi(e) : 1; }();
                if (queue) LG_DISPATCH(s, (k_lg_propQ<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, k)));
                else LG_DISPATCH(s, (k_lg_propN<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, k)));
            } else {
                LG_DISPATCH(s, (k_lg_prop<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, k)));
            }
        }
    }
    if (d.eb) {
        LG_DISPATCH(s, (void)DMAX; (k_lg_norm<ITEMS, 1><<<gtile, kBlock, 0, st>>>(d, d.T)));
        LG_DISPATCH(s, (void)DMAX; (k_lg_cdf<ITEMS, 1><<<gtile, kBlock, 0, st>>>(d, d.T)));
        k_lg_force_move<<<gone, 64, 0, st>>>(d);
        k_lg_path<<<dim3(gpath > (d.T + 64) / 64 ? gpath : (d.T + 64) / 64, d.C), 64, 0, st>>>(d, 1);
    } else {
        LG_DISPATCH(s, (void)DMAX; (k_lg_norm<ITEMS, 2><<<gtile, kBlock, 0, st>>>(d, d.T)));
        LG_DISPATCH(s, (void)DMAX; (k_lg_cdf<ITEMS, 2><<<gtile, kBlock, 0, st>>>(d, d.T)));
        k_lg_backscan<<<gone, 256, 0, st>>>(d);
    }
    {
        const int64_t n = (int64_t)d.N * d.du;
        int g = (int)((n + 255) / 256);
        g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
        k_lg_export<<<dim3(g, d.C), 256, 0, st>>>(d);
    }
    if (chain) {
        k_lg_advance<<<gone, 256, 0, st>>>(d);
        k_lg_advance_key<<<1, 64, 0, st>>>(d);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("sweep launch: ") + hipGetErrorString(e));
    return FBSMI_OK;
}

int get_graph(fbsmi_lg_sweep* s, int chain, hipGraphExec_t* out) {
    hipGraphExec_t& slot = chain ? s->graph_chain : s->graph_single;
    if (!slot) {
        hipGraph_t g = nullptr;
        FBSMI_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
        int rc = enqueue_sweep(s, s->stream, chain);
        hipError_t e = hipStreamEndCapture(s->stream, &g);
        if (rc) {
            if (g) hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        FBSMI_HIP_TRY(hipGraphInstantiate(&slot, g, nullptr, nullptr, 0));
        FBSMI_HIP_TRY(hipGraphDestroy(g));
    }
    *out = slot;
    return FBSMI_OK;
}

int run_sweep(fbsmi_lg_sweep* s, int chain, int use_graph) {
    if (use_graph && !s->profile) {
        hipGraphExec_t g;
        int rc = get_graph(s, chain, &g);
        if (rc) return rc;
        FBSMI_HIP_TRY(hipGraphLaunch(g, s->stream));
        return FBSMI_OK;
    }
    return enqueue_sweep(s, s->stream, chain);
}

int collect_profile(fbsmi_lg_sweep* s) {
    if (!s->profile) return FBSMI_OK;
    FBSMI_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int w = 0; w < kNumProfKernels; ++w) {
        auto& v = s->prof_ev[w];
        for (size_t i = 0; i + 1 < v.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, v[i], v[i + 1]) == hipSuccess) {
                s->prof_us[w] += (double)ms * 1000.0;
                s->prof_n[w] += 1;
            }
        }
        for (auto e : v) hipEventDestroy(e);
        v.clear();
    }
    return FBSMI_OK;
}

}  // namespace

extern "C" {

int fbsmi_lg_sweep_create(const fbsmi_lg_model* m, int32_t nparticles, int explicit_backward, int explicit_final,
                          int store_path, int32_t nchains, fbsmi_lg_sweep** out) {
    if (!m || !out || nparticles < 1 || m->du < 1 || m->dv < 1 || m->T < 1 || nchains < 1 || nchains > 65535)
        return fail(FBSMI_ERR_ARG, "lg_sweep_create: bad arguments");
    if (!m->G || !m->g || !m->sd || !m->lognorm || !m->F || !m->sqQ)
        return fail(FBSMI_ERR_ARG, "lg_sweep_create: null model table");
    const int D = m->du + m->dv;
    const bool wide = m->du > 16 || m->dv > 16;
    if (wide && (m->du > 128 || m->dv > 128))
        return fail(FBSMI_ERR_UNSUPPORTED, "lg_sweep: du, dv > 128 are not supported by the fused sweep");

    if (wide && (explicit_final ? nparticles + 1 : nparticles) > 131072)
        return fail(FBSMI_ERR_UNSUPPORTED, "lg_sweep: du, dv > 16 with more than 131072 particles is not supported");
    if (!explicit_backward && !store_path)
        return fail(FBSMI_ERR_ARG, "lg_sweep_create: explicit_backward=0 needs store_path (As, uss)");
    fbsmi_lg_sweep* s = new (std::nothrow) fbsmi_lg_sweep();
    if (!s) return fail(FBSMI_ERR_ARG, "out of host memory");
    LgDev& d = s->d;
    d.C = nchains;
    d.Ctot = nchains;
    d.c0 = 0;
    d.nparticles = nparticles;
    d.N = explicit_final ? nparticles + 1 : nparticles;
    d.du = m->du;
    d.dv = m->dv;
    d.D = D;
    d.T = m->T;
    d.eb = explicit_backward ? 1 : 0;
    d.ef = explicit_final ? 1 : 0;
    d.store = store_path ? 1 : 0;
    d.dt = m->dt;
    d.lw_init = (float)(-log((double)nparticles));
    d.G = m->G; d.g = m->g; d.sd = m->sd; d.lognorm = m->lognorm; d.F = m->F; d.sqQ = m->sqQ;
    d.levels = bisect_levels(d.N);
    d.wide = wide ? 1 : 0;
    d.lpw = nullptr;
    if (const char* dm = getenv("FBSMI_DEBUG_STEP_MASK")) s->debug_mask = atoi(dm);
    if (const char* gp = getenv("FBSMI_GENERIC_PROP")) s->generic_prop = atoi(gp) != 0;
    if (const char* sl = getenv("FBSMI_STEP_LAUNCHES")) s->step_launches = atoi(sl) != 0;
    if (const char* sp = getenv("FBSMI_TWO_SLOT_PROP")) s->two_slot_prop = atoi(sp) != 0 ? 1 : 0;
    if (const char* tp = getenv("FBSMI_TREE_STEP")) s->tree_step = atoi(tp) != 0;
    if (const char* th = getenv("FBSMI_TREE_HALVES")) s->tree_halves = atoi(th) == 1 ? 1 : (atoi(th) == 4 ? 4 : 2);
    s->items = fbsmi_tile_items(d.N);  // one workgroup = one tile of the two-level logsumexp (include/fbsmi_math.h)
    const int maxd = m->du > m->dv ? m->du : m->dv;
    s->dmax = maxd <= 1 ? 1 : (maxd <= 2 ? 2 : (maxd <= 4 ? 4 : 16));   // du = dv = 1 (BASELINE configs 1, 2) has its own instantiation
    const int tile = kBlock * s->items;
    d.nb = (d.N + tile - 1) / tile;
    if (d.nb > kMaxNbSweep) {
        delete s;
        return fail(FBSMI_ERR_UNSUPPORTED, "lg_sweep: more than 4M particles per device not supported yet");
    }
    if (store_path) {
        const double bytes = (((double)d.T * d.N * 4.0) + ((double)(d.T + 1) * d.N * (d.du + 1) * 4.0)) * nchains;
        if (bytes > 200e9) {
            delete s;
            return fail(FBSMI_ERR_UNSUPPORTED, "lg_sweep: path storage (T,N,du) does not fit device memory");
        }
    }
    const size_t N = d.N, T = d.T, C = nchains;
    int rc = 0;
    rc |= slab_request(s, &d.key, 2);
    rc |= slab_request(s, &d.keys, 2 * C);
    rc |= slab_request(s, &d.x0, C * d.du);
    rc |= slab_request(s, &d.y0, d.dv);
    rc |= slab_request(s, &d.bs, C * (T + 1));
    rc |= slab_request(s, &d.keytab, C * 8 * T);
    rc |= slab_request(s, &d.misc, C * 16);
    rc |= slab_request(s, &d.xi1, C * T * D);
    rc |= slab_request(s, &d.xi2, C * T * D);
    rc |= slab_request(s, &d.path, 1);
    rc |= slab_request(s, &d.us_star, C * (T + 1) * d.du);
    rc |= slab_request(s, &d.vs, C * (T + 1) * d.dv);
    rc |= slab_request(s, &d.u0, C * N * d.du);
    rc |= slab_request(s, &d.u1, C * N * d.du);
    rc |= slab_request(s, &d.lw, C * N);
    rc |= slab_request(s, &d.lwn, C * N);
    rc |= slab_request(s, &d.w, C * N);
    rc |= slab_request(s, &d.cdf, C * N);
    rc |= slab_request(s, &d.cdfJ, C * N);
    rc |= slab_request(s, &d.bmax, C * d.nb);
    rc |= slab_request(s, &d.bsumexp, C * d.nb);
    d.anc = nullptr;
    d.xiw = nullptr;
    d.uw = nullptr;
    if (wide) {
        rc |= slab_request(s, &d.lpw, C * (size_t)((d.dv + 3) & ~3) * N);
        rc |= slab_request(s, &d.anc, C * N);
        rc |= slab_request(s, &d.xiw, 2 * C * N * d.du);
        static const int upre = [] { const char* e = getenv("FBSMI_WIDE_UPRE"); return e ? atoi(e) : 1; }();
        if (upre && N <= kBlock) rc |= slab_request(s, &d.uw, 4 * C * (size_t)N);
    }
    d.hpW = d.hpJ = nullptr;
    d.hp_map = nullptr;
    int32_t* hp_map_dev = nullptr;
    d.lh_w = d.lh_j = 0;
    if (!wide) {   // compact bisection heaps: written by k_lg_cdf itself (one slot per thread, through hp_map) or by k_lg_heaps
        int fl = 0;
        while ((2ll << fl) <= (long long)d.N) ++fl;   // floor(log2 N)
        d.lh_w = fl < kHeapLevelsW ? fl : kHeapLevelsW;
        d.lh_j = fl < kHeapLevelsJ ? fl : kHeapLevelsJ;
        if (s->items == 1) rc |= slab_request(s, &hp_map_dev, (size_t)N);
        rc |= slab_request(s, &d.hpW, C * (size_t)kHeapSizeW);
        rc |= slab_request(s, &d.hpJ, C * (size_t)kHeapSizeJ);
    }
    d.trW = d.trWtop = nullptr;
    d.wfirst = nullptr;
    const bool pow2 = (d.N & (d.N - 1)) == 0 && d.nb >= 2 && d.nb <= kBlock;
    // N = 2^k + 1 (explicit_final on a power-of-two ensemble): the same step over the first 2^k slots' tree plus an extra
    // one-slot tile (tree_build)
    const bool pow2p1 = d.N > 2 && ((d.N - 1) & (d.N - 2)) == 0 && d.nb - 1 >= 2 && d.nb - 1 <= kBlock &&
                        !(getenv("FBSMI_TREE_PLUS1") && atoi(getenv("FBSMI_TREE_PLUS1")) == 0);
    d.plus1 = 0;
    if (s->items == 1 && !wide && (pow2 || pow2p1)) {
        d.plus1 = pow2 ? 0 : 1;
        rc |= slab_request(s, &d.trW, C * (size_t)kTreeNodes * d.nb);
        rc |= slab_request(s, &d.trWtop, C * (size_t)kMidN * d.nb);
        rc |= slab_request(s, &d.wfirst, C * (size_t)d.nb);
        // few workgroups per launch (BASELINE config 1: 4 tiles; up to 64 workgroups): keep the step on one XCD.  Only with the default kernel
        // choices (one tile per workgroup, one slot per thread), which is what such sizes get.  FBSMI_PIN=0 turns it off.
        const char* pe = getenv("FBSMI_PIN");
        const char* pm = getenv("FBSMI_PIN_MAX");   // (diagnostic: workgroups per launch up to which the step is pinned)
        const int64_t pin_max = pm ? atoi(pm) : 64;   // two workgroups per CU of the XCD: measured better up to there, worse beyond
        d.pin = (!d.plus1 && s->tree_step && !s->generic_prop && s->tree_halves < 0 && s->two_slot_prop < 0 && (int64_t)d.nb * C <= pin_max &&
                 (int64_t)d.nb * C < 2 * 256 && !(pe && atoi(pe) == 0)) ? 1 : 0;
        if (d.pin && !xcd_round_robin_holds()) d.pin = 0;   // the dispatch order the pinning rests on is not what this machine does
    }
    rc |= slab_request(s, &d.bsumw, C * d.nb);
    rc |= slab_request(s, &d.bsumJ, C * d.nb);
    rc |= slab_request(s, &d.scal, C * 16);
    rc |= slab_request(s, &d.usT, C * N * d.du);
    rc |= slab_request(s, &d.x0n, C * d.du);
    rc |= slab_request(s, &d.usn, C * (T + 1) * d.du);
    rc |= slab_request(s, &d.bsn, C * (T + 1));
    rc |= slab_request(s, &d.acc, C * (T + 1));
    rc |= slab_request(s, &d.x0s_slot, 1);
    rc |= slab_request(s, &d.counter, 1);
    rc |= slab_request(s, &d.dbg, 64);
    if (store_path) {
        rc |= slab_request(s, &d.As, C * T * N);
        rc |= slab_request(s, &d.uss, C * (T + 1) * N * d.du);
        rc |= slab_request(s, &d.lwss, C * (T + 1) * N);
    }
    rc |= slab_commit(s);
    if (!rc && hp_map_dev) {
        // Down to depth floor(log2 N) every node of the implicit bisection tree is at least two wide, so an
        // element is the midpoint of at most one node there: walk the tree once, on the host.
        std::vector<int32_t> map((size_t)d.N, 0);
        struct Node { int lo, hi, t, dep; };
        std::vector<Node> stack{{0, d.N, 1, 0}};
        while (!stack.empty()) {
            const Node n = stack.back();
            stack.pop_back();
            if (n.dep >= d.lh_w) continue;
            const int mid = (n.lo + n.hi) >> 1;
            map[mid] = n.t | (n.dep << 16);
            stack.push_back({n.lo, mid, 2 * n.t, n.dep + 1});
            stack.push_back({mid, n.hi, 2 * n.t + 1, n.dep + 1});
        }
        if (hipMemcpy(hp_map_dev, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
        d.hp_map = hp_map_dev;
    }
    if (rc) {
        fbsmi_lg_sweep_destroy(s);
        return FBSMI_ERR_HIP;
    }
    (void)xcd_round_robin_holds();   // the once-per-process placement probe must not run inside a later stream capture
    if ((s->stream = pool_stream(0)) == nullptr ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming) != hipSuccess) {
        fbsmi_lg_sweep_destroy(s);
        return fail(FBSMI_ERR_HIP, "lg_sweep_create: stream/event creation failed");
    }
    if (wide) {
        // The attribute belongs to the function, not to the handle: always ask for the largest tile pair any model can
        // need (D = 256), or a later handle with a smaller model would lower the limit under an earlier one.
        const int lds = (int)wide_lds_bytes(wide_plane_row(256));
        hipError_t e = hipFuncSetAttribute((const void*)k_lgw_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm_fat<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_lgw_gemm_fat<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            fbsmi_lg_sweep_destroy(s);
            return fail(FBSMI_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        }
    }
    *out = s;
    return FBSMI_OK;
}

void fbsmi_lg_sweep_destroy(fbsmi_lg_sweep* s) {
    if (!s) return;
    if (s->stream) hipStreamSynchronize(s->stream);
    if (s->graph_single) hipGraphExecDestroy(s->graph_single);
    if (s->graph_chain) hipGraphExecDestroy(s->graph_chain);
    for (int w = 0; w < kNumProfKernels; ++w)
        for (auto e : s->prof_ev[w]) hipEventDestroy(e);
    if (s->ev_in) hipEventDestroy(s->ev_in);
    if (s->ev_out) hipEventDestroy(s->ev_out);
    // (the stream belongs to the pool)
    for (void* p : s->allocs) hipFree(p);
    delete s;
}

int fbsmi_lg_gibbs_sweep(fbsmi_lg_sweep* s, const uint32_t* keys, const float* x0, const float* y0,
                         const int32_t* bs_star, float* x0_next, float* us_star_next, int32_t* bs_next,
                         uint8_t* acc, int use_graph, void* stream) {
    if (!s || !keys || !x0 || !y0 || !bs_star) return fail(FBSMI_ERR_ARG, "lg_gibbs_sweep: null input");
    hipStream_t ust = (hipStream_t)stream;
    const LgDev& d = s->d;
    const size_t C = d.C, T1 = d.T + 1;
    FBSMI_HIP_TRY(hipEventRecord(s->ev_in, ust));
    FBSMI_HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_in, 0));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.keys, keys, C * 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.x0, x0, C * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.y0, y0, d.dv * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.bs, bs_star, C * T1 * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    int rc = run_sweep(s, 0, use_graph);
    if (rc) return rc;
    if (x0_next) FBSMI_HIP_TRY(hipMemcpyAsync(x0_next, d.x0n, C * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    if (us_star_next)
        FBSMI_HIP_TRY(hipMemcpyAsync(us_star_next, d.usn, C * T1 * d.du * sizeof(float), hipMemcpyDeviceToDevice,
                                     s->stream));
    if (bs_next) FBSMI_HIP_TRY(hipMemcpyAsync(bs_next, d.bsn, C * T1 * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    if (acc) FBSMI_HIP_TRY(hipMemcpyAsync(acc, d.acc, C * T1, hipMemcpyDeviceToDevice, s->stream));
    rc = collect_profile(s);
    if (rc) return rc;
    FBSMI_HIP_TRY(hipEventRecord(s->ev_out, s->stream));
    FBSMI_HIP_TRY(hipStreamWaitEvent(ust, s->ev_out, 0));
    return FBSMI_OK;
}

namespace {
// inputs of a chain call -> the handle's buffers, on its own stream (x0 / bs_star / x0s: this handle's chains only)
int chain_begin(fbsmi_lg_sweep* s, const uint32_t* key, const float* x0, const float* y0, const int32_t* bs_star, float* x0s,
                hipStream_t ust) {
    const LgDev& d = s->d;
    const size_t C = d.C, T1 = d.T + 1;
    FBSMI_HIP_TRY(hipEventRecord(s->ev_in, ust));
    FBSMI_HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_in, 0));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.key, key, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.x0, x0, C * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.y0, y0, d.dv * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.bs, bs_star, C * T1 * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemsetAsync(d.counter, 0, sizeof(int32_t), s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.x0s_slot, &x0s, sizeof(float*), hipMemcpyHostToDevice, s->stream));
    // the slot copy reads a host stack variable: make sure it has landed before we return
    FBSMI_HIP_TRY(hipStreamSynchronize(s->stream));
    return FBSMI_OK;
}

int chain_end(fbsmi_lg_sweep* s, uint32_t* key, float* x0, int32_t* bs_star, hipStream_t ust) {
    const LgDev& d = s->d;
    const size_t C = d.C, T1 = d.T + 1;
    if (key) FBSMI_HIP_TRY(hipMemcpyAsync(key, d.key, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(x0, d.x0, C * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(bs_star, d.bs, C * T1 * sizeof(int32_t), hipMemcpyDeviceToDevice, s->stream));
    int rc = collect_profile(s);
    if (rc) return rc;
    FBSMI_HIP_TRY(hipEventRecord(s->ev_out, s->stream));
    FBSMI_HIP_TRY(hipStreamWaitEvent(ust, s->ev_out, 0));
    return FBSMI_OK;
}
}  // namespace

int fbsmi_lg_gibbs_chain(fbsmi_lg_sweep* s, uint32_t* key, float* x0, const float* y0, int32_t* bs_star,
                         int32_t nsweeps, float* x0s, int use_graph, void* stream) {
    if (!s || !key || !x0 || !y0 || !bs_star || nsweeps < 0) return fail(FBSMI_ERR_ARG, "lg_gibbs_chain: bad arguments");
    hipStream_t ust = (hipStream_t)stream;
    int rc = chain_begin(s, key, x0, y0, bs_star, x0s, ust);
    if (rc) return rc;
    for (int i = 0; i < nsweeps; ++i) {
        rc = run_sweep(s, 1, use_graph);
        if (rc) return rc;
    }
    return chain_end(s, key, x0, bs_star, ust);
}

int fbsmi_lg_sweep_set_group(fbsmi_lg_sweep* s, int32_t nchains_total, int32_t first_chain) {
    if (!s || nchains_total < s->d.C || first_chain < 0 || first_chain + s->d.C > nchains_total)
        return fail(FBSMI_ERR_ARG, "lg_sweep_set_group: bad arguments");
    if (s->graph_chain) return fail(FBSMI_ERR_ARG, "lg_sweep_set_group: call it before the handle's first chain sweep");
    s->d.Ctot = nchains_total;
    s->d.c0 = first_chain;
    if (hipStream_t st = pool_stream(first_chain / s->d.C)) s->stream = st;   // group g of the batch: pool stream g
    return FBSMI_OK;
}

int fbsmi_lg_gibbs_chain_groups(fbsmi_lg_sweep* const* groups, int32_t ngroups, uint32_t* key, float* x0, const float* y0,
                                int32_t* bs_star, int32_t nsweeps, float* x0s, int use_graph, void* stream) {
    if (!groups || ngroups < 1 || !key || !x0 || !y0 || !bs_star || nsweeps < 0)
        return fail(FBSMI_ERR_ARG, "lg_gibbs_chain_groups: bad arguments");
    hipStream_t ust = (hipStream_t)stream;
    int expect = 0;
    for (int g = 0; g < ngroups; ++g) {
        fbsmi_lg_sweep* s = groups[g];
        if (!s || s->d.c0 != expect || s->d.Ctot != groups[0]->d.Ctot || s->d.T != groups[0]->d.T || s->d.du != groups[0]->d.du)
            return fail(FBSMI_ERR_ARG, "lg_gibbs_chain_groups: the handles must cover chains 0 .. Ctot-1 in order (set_group)");
        expect += s->d.C;
    }
    if (expect != groups[0]->d.Ctot) return fail(FBSMI_ERR_ARG, "lg_gibbs_chain_groups: the handles do not cover the batch");
    // group g runs on pool stream g (set_group's first_chain / C is the same number for groups of equal size; groups of unequal
    // sizes could collide on it).  Only before a handle's graph exists: a captured graph replays on the stream it was captured on.
    for (int g = 0; g < ngroups; ++g)
        if (!groups[g]->graph_chain)
            if (hipStream_t st = pool_stream(g)) groups[g]->stream = st;
    for (int g = 0; g < ngroups; ++g) {
        fbsmi_lg_sweep* s = groups[g];
        const size_t c0 = s->d.c0;
        int rc = chain_begin(s, key, x0 + c0 * s->d.du, y0, bs_star + c0 * (s->d.T + 1), x0s, ust);
        if (rc) return rc;
    }
    // sweeps outermost: every group's stream always has work queued, and the groups' launches interleave on the GPU
    for (int i = 0; i < nsweeps; ++i)
        for (int g = 0; g < ngroups; ++g) {
            int rc = run_sweep(groups[g], 1, use_graph);
            if (rc) return rc;
        }
    for (int g = 0; g < ngroups; ++g) {
        fbsmi_lg_sweep* s = groups[g];
        const size_t c0 = s->d.c0;
        int rc = chain_end(s, g == 0 ? key : nullptr, x0 + c0 * s->d.du, bs_star + c0 * (s->d.T + 1), ust);
        if (rc) return rc;
    }
    return FBSMI_OK;
}

int fbsmi_lg_sweep_view(fbsmi_lg_sweep* s, int which, void* dst, int64_t* count, void* stream) {
    if (!s) return fail(FBSMI_ERR_ARG, "lg_sweep_view: null handle");
    const LgDev& d = s->d;
    const void* src = nullptr;
    int64_t n = 0;
    size_t esz = 4;
    switch (which) {
        case 0: src = d.usT; n = (int64_t)d.C * d.N * d.du; break;
        case 1: src = d.lwn; n = (int64_t)d.C * d.N; break;
        case 2: src = d.As; n = (int64_t)d.C * d.T * d.N; break;
        case 3: src = d.uss; n = (int64_t)d.C * (d.T + 1) * d.N * d.du; break;
        case 4: src = d.lwss; n = (int64_t)d.C * (d.T + 1) * d.N; break;
        case 5: src = d.us_star; n = (int64_t)d.C * (d.T + 1) * d.du; break;
        case 6: src = d.vs; n = (int64_t)d.C * (d.T + 1) * d.dv; break;
        case 7: src = d.dbg; n = 128; break;  // 64 x uint64 as 32-bit words (diagnostic build)
        case 8: src = d.cdfJ; n = (int64_t)d.C * d.N; break;   // (diagnostic build: per-workgroup entry / exit stamps of the last step)
        case 9: src = d.xiw ? d.xiw + (size_t)d.N * d.du : nullptr; n = 16384; break;   // (diagnostic build: the drift kernel's workgroups)
        default: return fail(FBSMI_ERR_ARG, "lg_sweep_view: unknown view");
    }
    if (!src) n = 0;
    if (count) *count = n;
    if (dst && n > 0) {
        hipStream_t ust = (hipStream_t)stream;
        FBSMI_HIP_TRY(hipStreamSynchronize(s->stream));
        FBSMI_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * esz, hipMemcpyDeviceToDevice, ust));
    }
    return FBSMI_OK;
}

// ---- fused particle filters ------------------------------------------------------------------
struct fbsmi_lg_filter {
    fbsmi_lg_sweep* core = nullptr;  // buffers, stream, events
    float* u0s = nullptr;            // [C][n][du] staging of the initial particles
    hipGraphExec_t graph = nullptr;
};

namespace {

int enqueue_filter(fbsmi_lg_filter* f, hipStream_t st) {
    fbsmi_lg_sweep* s = f->core;
    const LgDev& d = s->d;
    const dim3 gone(1, d.C), gtile(d.nb, d.C);
    k_filt_keys<<<gone, kBlock, 0, st>>>(d);
    if (!d.wide && d.N <= kBlock && !s->step_launches) {
        LG_DISPATCH(s, (void)ITEMS; (k_filt_sweep1<DMAX><<<gone, kBlock, 0, st>>>(d, f->u0s)));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("filter launch: ") + hipGetErrorString(e));
        return FBSMI_OK;
    }
    if (d.wide) {
        // N <= 256: a launch = [filter prologue +] drift product
        const int nrt = (d.D + kWideTile - 1) / kWideTile, Kp = (d.D + 15) / 16 * 16, S = wide_plane_row(Kp);
        const size_t lds = wide_lds_bytes(S);
        const int nst = (d.N + kWideTile - 1) / kWideTile;
        const int u_tiles = (d.du + kWideTile - 1) / kWideTile;   // row tiles holding rows < du
        const int v_tile0 = d.du / kWideTile;                     // first row tile holding rows >= du
        k_lgwf_init<<<dim3(8, d.C), kBlock, 0, st>>>(d, f->u0s);
        if (d.N > kBlock) {
            // several logsumexp tiles: the prologue is its own launches (row sums + partials, normalise, cumsum,
            // ancestors), the drift product takes the ancestors from d.anc
            const dim3 gall(nst * nrt, d.C), gu(nst * u_tiles, d.C), gv(nst * (nrt - v_tile0), d.C), gg(64, d.C);
            auto resample = [&](int kres) {
                k_lgw_lse<<<gtile, kBlock, 0, st>>>(d);
                k_filt_norm<1><<<gtile, kBlock, 0, st>>>(d);
                k_lg_cdf<1, 2><<<gtile, kBlock, 0, st>>>(d, d.T);
                k_lgwf_anc<<<gtile, kBlock, 0, st>>>(d, kres);
            };
            if (d.flow == 0) {
                k_lgw_gemm<3><<<gall, kBlock, lds, st>>>(d, 0, 0, nrt, Kp, S, 3, 0);
                for (int k = 1; k < d.T; ++k) {
                    resample(k - 1);
                    k_lgw_gemm<4><<<gall, kBlock, lds, st>>>(d, k, 0, nrt, Kp, S, 3, 0);
                }
                resample(d.T - 1);
                k_lgwf_gather<<<gg, kBlock, 0, st>>>(d);
            } else {
                k_lgw_gemm<3><<<gv, kBlock, lds, st>>>(d, 0, v_tile0, nrt - v_tile0, Kp, S, 2, 0);
                for (int k = 0; k < d.T; ++k) {
                    resample(k);
                    k_lgw_gemm<4><<<gu, kBlock, lds, st>>>(d, k, 0, u_tiles, Kp, S, 1, 0);
                    if (k + 1 < d.T) k_lgw_gemm<3><<<gv, kBlock, lds, st>>>(d, k + 1, v_tile0, nrt - v_tile0, Kp, S, 2, 0);
                }
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("filter launch: ") + hipGetErrorString(e));
            return FBSMI_OK;
        }
        // one-tile ensembles: a step's launches are a few dozen workgroups -- pinned to one XCD (bit 8 of `emit`, grids 8x wide)
        static const int pin_on = [] { const char* e = getenv("FBSMI_WIDE_PIN"); return (e ? atoi(e) : 1) && xcd_round_robin_holds(); }();
        const bool pinw = pin_on && d.C == 1 && nst * nrt <= 32;
        const int pe = pinw ? 0x100 : 0, pg = pinw ? 8 : 1;
        if (d.flow == 0) {
            for (int k = 0; k < d.T; ++k) {
                if (k == 0) k_lgw_gemm<3><<<dim3(nst * nrt * pg, d.C), kBlock, lds, st>>>(d, k, 0, nrt, Kp, S, 3 | pe, 0);
                else k_lgw_gemm<2><<<dim3(nst * nrt * pg, d.C), kBlock, lds, st>>>(d, k, 0, nrt, Kp, S, 3 | pe, k - 1);
            }
            k_lgwf_final<<<gone, kBlock, 0, st>>>(d);
        } else {
            // weight the current particles (rows >= du, no resampling), then resample + propagate (rows < du)
            k_lgw_gemm<3><<<dim3(nst * (nrt - v_tile0) * pg, d.C), kBlock, lds, st>>>(d, 0, v_tile0, nrt - v_tile0, Kp, S, 2 | pe, 0);
            for (int k = 0; k < d.T; ++k) {
                k_lgw_gemm<2><<<dim3(nst * u_tiles * pg, d.C), kBlock, lds, st>>>(d, k, 0, u_tiles, Kp, S, 1 | pe, k);
                if (k + 1 < d.T)
                    k_lgw_gemm<3><<<dim3(nst * (nrt - v_tile0) * pg, d.C), kBlock, lds, st>>>(d, k + 1, v_tile0, nrt - v_tile0,
                                                                                               Kp, S, 2 | pe, 0);
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("filter launch: ") + hipGetErrorString(e));
        return FBSMI_OK;
    }
    LG_DISPATCH(s, (k_filt_init<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, f->u0s)));
    // N a power of two with 2..256 tiles: the searches walk the summation tree, a step is two launches (no cdf)
    const bool tree = s->tree_step && d.trW && s->items == 1 && !d.plus1;
    if (tree) {
        const dim3 gpin(gtile.x * (d.pin ? 8 : 1), d.C);   // (pinned to one XCD when the launches are small: LgDev.pin)
        for (int k = 0; k < d.T; ++k) {
            if (d.flow == 0)
                LG_DISPATCH(s, (void)ITEMS; (k_filt_prop1t<DMAX><<<gpin, kBlock, 0, st>>>(d, k, k > 0, k > 0 ? k - 1 : 0, 1, 1)));
            k_filt_norm<1, true><<<gpin, kBlock, 0, st>>>(d);
            if (d.flow == 1)
                LG_DISPATCH(s, (void)ITEMS; (k_filt_prop1t<DMAX><<<gpin, kBlock, 0, st>>>(d, k, 1, k, k + 1 < d.T ? 2 : 0, 1)));
        }
        if (d.flow == 0)
            LG_DISPATCH(s, (void)ITEMS; (k_filt_prop1t<DMAX><<<gpin, kBlock, 0, st>>>(d, d.T, 1, d.T - 1, 0, 0)));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("filter launch: ") + hipGetErrorString(e));
        return FBSMI_OK;
    }
    for (int k = 0; k < d.T; ++k) {
        if (d.flow == 0)
            LG_DISPATCH(s, (k_filt_prop<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, k, k > 0, k > 0 ? k - 1 : 0, 1, 1)));
        LG_DISPATCH(s, (void)DMAX; (k_filt_norm<ITEMS><<<gtile, kBlock, 0, st>>>(d)));
        LG_DISPATCH(s, (void)DMAX; (k_lg_cdf<ITEMS, 2><<<gtile, kBlock, 0, st>>>(d, d.T)));
        if (d.flow == 1)
            LG_DISPATCH(s, (k_filt_prop<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, k, 1, k, k + 1 < d.T ? 2 : 0, 1)));
    }
    if (d.flow == 0)
        LG_DISPATCH(s, (k_filt_prop<ITEMS, DMAX><<<gtile, kBlock, 0, st>>>(d, d.T, 1, d.T - 1, 0, 0)));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("filter launch: ") + hipGetErrorString(e));
    return FBSMI_OK;
}

}  // namespace

int fbsmi_lg_filter_create(const fbsmi_lg_model* m, int32_t nparticles, int flow, int resampling, int store_path,
                           int32_t nchains, fbsmi_lg_filter** out) {
    if (!out || flow < 0 || flow > 1 || resampling < 0 || resampling > 1)
        return fail(FBSMI_ERR_ARG, "lg_filter_create: flow must be 0|1 and resampling 0 (stratified) | 1 (systematic)");
    if (flow == 1 && store_path) return fail(FBSMI_ERR_ARG, "lg_filter_create: pmcmc_filter_step keeps no path");

    fbsmi_lg_sweep* core = nullptr;
    int rc = fbsmi_lg_sweep_create(m, nparticles, 1, 0, store_path, nchains, &core);
    if (rc) return rc;
    fbsmi_lg_filter* f = new (std::nothrow) fbsmi_lg_filter();
    if (!f) {
        fbsmi_lg_sweep_destroy(core);
        return fail(FBSMI_ERR_ARG, "out of host memory");
    }
    f->core = core;
    LgDev& d = core->d;
    d.flow = flow;
    d.systematic = resampling;
    d.logn = (float)log((double)nparticles);
    if (dev_alloc(core, &d.ell, (size_t)d.C) || dev_alloc(core, &f->u0s, (size_t)d.C * d.N * d.du)) {
        fbsmi_lg_sweep_destroy(core);
        delete f;
        return FBSMI_ERR_HIP;
    }
    *out = f;
    return FBSMI_OK;
}

void fbsmi_lg_filter_destroy(fbsmi_lg_filter* f) {
    if (!f) return;
    if (f->core && f->core->stream) hipStreamSynchronize(f->core->stream);
    if (f->graph) hipGraphExecDestroy(f->graph);
    fbsmi_lg_sweep_destroy(f->core);
    delete f;
}

int fbsmi_lg_filter_run(fbsmi_lg_filter* f, const uint32_t* keys, const float* vs, const float* u0s, float* uT,
                        float* loglik, float* path, int use_graph, void* stream) {
    if (!f || !keys || !vs || !u0s) return fail(FBSMI_ERR_ARG, "lg_filter_run: null input");
    fbsmi_lg_sweep* s = f->core;
    const LgDev& d = s->d;
    if (path && !d.uss) return fail(FBSMI_ERR_ARG, "lg_filter_run: path requested but the filter was created without store_path");
    hipStream_t ust = (hipStream_t)stream;
    const size_t C = d.C, T1 = d.T + 1;
    FBSMI_HIP_TRY(hipEventRecord(s->ev_in, ust));
    FBSMI_HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_in, 0));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.keys, keys, C * 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(d.vs, vs, C * T1 * d.dv * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipMemcpyAsync(f->u0s, u0s, C * d.N * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    if (use_graph) {
        if (!f->graph) {
            hipGraph_t g = nullptr;
            FBSMI_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
            int rc = enqueue_filter(f, s->stream);
            hipError_t e = hipStreamEndCapture(s->stream, &g);
            if (rc) return rc;
            if (e != hipSuccess) return fail(FBSMI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            FBSMI_HIP_TRY(hipGraphInstantiate(&f->graph, g, nullptr, nullptr, 0));
            FBSMI_HIP_TRY(hipGraphDestroy(g));
        }
        FBSMI_HIP_TRY(hipGraphLaunch(f->graph, s->stream));
    } else {
        int rc = enqueue_filter(f, s->stream);
        if (rc) return rc;
    }
    if (uT) FBSMI_HIP_TRY(hipMemcpyAsync(uT, d.usT, C * d.N * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    if (loglik) FBSMI_HIP_TRY(hipMemcpyAsync(loglik, d.ell, C * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    if (path)
        FBSMI_HIP_TRY(hipMemcpyAsync(path, d.uss, C * T1 * d.N * d.du * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    FBSMI_HIP_TRY(hipEventRecord(s->ev_out, s->stream));
    FBSMI_HIP_TRY(hipStreamWaitEvent(ust, s->ev_out, 0));
    return FBSMI_OK;
}

int fbsmi_lg_sweep_profile(fbsmi_lg_sweep* s, int enable) {
    if (!s) return fail(FBSMI_ERR_ARG, "lg_sweep_profile: null handle");
    s->profile = enable != 0;
    for (int w = 0; w < kNumProfKernels; ++w) {
        s->prof_us[w] = 0;
        s->prof_n[w] = 0;
    }
    return FBSMI_OK;
}

int fbsmi_lg_sweep_kernel_us(fbsmi_lg_sweep* s, int which, double* avg_us, int64_t* launches) {
    if (!s || which < 0 || which >= kNumProfKernels) return fail(FBSMI_ERR_ARG, "lg_sweep_kernel_us: bad arguments");
    if (avg_us) *avg_us = s->prof_n[which] ? s->prof_us[which] / (double)s->prof_n[which] : 0.0;
    if (launches) *launches = s->prof_n[which];
    return FBSMI_OK;
}

}  // extern "C"
