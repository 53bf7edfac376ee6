// fbsmi_em.hip -- the SMC step of score-network models (BASELINE configs 3-5) around ONE network
// evaluation per step: what experiments/imgs/inpainting.py:102-147 (and supr.py, sb_imgs/supr.py:80-127)
// spread over concat / unpack / drift / Euler-Maruyama / norm.logpdf / jnp.sum, as two kernels.
//
//   k_em_concat : img[r] = concat(us[A[r]], v_prev)            (csmc.py:140, fbs/data/images.py:355-363)
//   k_em_finish : unpack(net) -> reverse drift -> proposal with in-kernel jax.random.normal -> pin
//                 -> row-summed Gaussian log-density             (inpainting.py:102-147, csmc.py:143-145)
//
// This is the one place on the path where the per-particle state is kilobytes, so the bound is HBM
// bandwidth: per particle the finish kernel reads the network output once (4 D bytes), the ancestor's
// row once (4 du) and writes the new row once (4 du) and one log-weight.  Its workgroups have two roles,
// interleaved over the grid so that both kinds are resident on every CU at the same time:
//   * proposal groups walk the FLAT index space of the (rows, du) draw.  jax's random_bits puts elements
//     i and i + n/2 of a draw on the two output words of one Threefry call, so a thread that owns
//     elements i..i+3 also owns i+n/2..i+n/2+3: eight normals out of four block-cipher calls.  These
//     groups are bound by vector-instruction issue (Threefry-2x32/20 + the erf_inv expansion, ~200
//     instructions per normal);
//   * log-density groups own one particle row: every wave reduces 256 consecutive observed elements at a
//     time in the canonical pairwise order (4 in the lane, 6 butterfly levels), the segment sums meet in
//     LDS.  These groups are bound by the memory system.
// Reads of the network output go through the mask's offset tables (runs of s*c contiguous floats for an
// s x s inpainting rectangle): 16-byte loads wherever a lane's four offsets are consecutive.
// gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "../../include/fbsmi.h"
#include "fbsmi_device.h"
#include "fbsmi_host.h"

namespace fbsmi {

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));           // 16-byte load, dword aligned
typedef unsigned short us4u __attribute__((ext_vector_type(4), aligned(2)));  // four bf16, 2-byte aligned

// FBSMI_EM_PROBE (diagnostic builds, tools/build_variants.sh; results are WRONG): 1 no erf_inv, 2 no Threefry and no
// erf_inv, 3 log-density role without its arithmetic, 4 proposal role without its global loads, 5 (k_em_rows) no log-density
#ifndef FBSMI_EM_PROBE
#define FBSMI_EM_PROBE 0
#endif

#ifndef FBSMI_EM_WAVES
#define FBSMI_EM_WAVES 8  // waves per SIMD the finish kernel is compiled for (56 registers used)
#endif

struct EmArgs {
    const float* us;
    const int32_t* A;
    const int32_t* net_A;  // row of `net` that belongs to local row r (NULL: r)
    const void* net;
    const float* v;        // target of the log-density (observed part: v; transition_logpdf: u)
    const float* v_prev;
    const float* pin_value;
    float* us_new;
    float* lw;
    const int32_t* u_off;
    const int32_t* v_off;
    int32_t du, dv, D;
    float cx, cs, dt, sd;
    uint32_t k0, k1;
    uint32_t ntot_el;  // n_total * du: size of the flat draw
    uint32_t half;     // (ntot_el + 1) / 2
    uint32_t first_el; // row0 * du
    uint32_t nloc_el;  // n * du
    int32_t n;
    int32_t pin_row;   // local row, -1: none
    int32_t nU, nV;    // proposal / log-density workgroups
    int32_t vfirst;    // the log-density workgroups (one long row each) take the first block indices, the short proposal ones fill in
};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return fbsmi_u2f((uint32_t)h << 16); }

// round to nearest even; NaN stays NaN
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    const uint32_t u = fbsmi_f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int NETDT>
__device__ __forceinline__ float net_at(const void* net, int64_t idx) {
    if (NETDT == 0) return ((const float*)net)[idx];
    return bf16_to_f32(((const unsigned short*)net)[idx]);
}

// Four network values at offsets o[0..3] of a row (row_base + o[k] inside the buffer, D floats per row).  Loads under
// a DIVERGENT branch cost a full drain of the wave's memory queue at the branch's end on this compiler (it parks an
// s_waitcnt vmcnt(0) where the loaded registers meet the other path's), so nothing here is predicated per lane: every
// lane issues one 16-byte load at min(o[0], D - 4), which is the four values whenever its offsets are consecutive
// (inside a run of the mask); if ANY lane of the wave has a group across two runs the whole wave also issues the four
// element loads (a wave-uniform branch) and each lane picks.  The address unit takes four lanes per cycle whatever the
// width, so a dword gather costs as much as a 16-byte one: wide loads are the point.
template <int NETDT>
__device__ __forceinline__ void net_at4(const void* net, int64_t row_base, int D, const int (&o)[4], bool lane_valid,
                                        float (&s)[4]) {
    const bool contig = o[3] - o[0] == 3;
    const int wb = o[0] < D - 4 ? o[0] : D - 4;   // == o[0] when contig
    float w[4];
    if (NETDT == 0) {
        const f4u t = *(const f4u*)((const float*)net + row_base + wb);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else {
        const us4u t = *(const us4u*)((const unsigned short*)net + row_base + wb);
        w[0] = bf16_to_f32(t.x); w[1] = bf16_to_f32(t.y); w[2] = bf16_to_f32(t.z); w[3] = bf16_to_f32(t.w);
    }
    if (__any(lane_valid && !contig)) {
        float nv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) nv[k] = net_at<NETDT>(net, row_base + o[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = contig ? w[k] : nv[k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = w[k];
    }
}

template <int MODE>
__device__ __forceinline__ float em_drift(float cx, float cs, float x, float s) {
    if (MODE == 1) return s;
    const float t1 = cx * x, t2 = cs * s;
    return t1 + t2;
}

// ------------------------------------------------------------------------------------------------
// proposal: four consecutive flat elements [e0, e0 + 4) of the global draw (those below `lim`), with
// their four random words.  VEC: du % 4 == 0 and e0 % 4 == 0, so the four sit in one row, 16-byte
// aligned.
// ------------------------------------------------------------------------------------------------
struct UGroup {  // operands of four consecutive elements of one row (VEC walk)
    uint32_t r, p;
    float x[4], s[4];
};

template <int NETDT>
__device__ __forceinline__ void em_u_load(const EmArgs& a, uint32_t e0, UGroup& g) {
    const uint32_t le = e0 - a.first_el;
    const uint32_t du = (uint32_t)a.du;
    g.r = le / du;
    g.p = le - g.r * du;
    if (FBSMI_EM_PROBE == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { g.x[k] = (float)g.p; g.s[k] = (float)g.r; }
        return;
    }
    const int64_t src = a.A ? (int64_t)a.A[g.r] : (int64_t)g.r;
    const int4 o4 = *(const int4*)(a.u_off + g.p);
    const float4 x4 = *(const float4*)(a.us + src * du + g.p);
    const int o[4] = {o4.x, o4.y, o4.z, o4.w};
    net_at4<NETDT>(a.net, (int64_t)(a.net_A ? a.net_A[g.r] : (int32_t)g.r) * a.D, a.D, o, true, g.s);
    g.x[0] = x4.x; g.x[1] = x4.y; g.x[2] = x4.z; g.x[3] = x4.w;
}

template <int NETDT, int MODE>
__device__ __forceinline__ void em_u_store(const EmArgs& a, const UGroup& g, const uint32_t (&bits)[4]) {
    float y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float z = (FBSMI_EM_PROBE == 1 || FBSMI_EM_PROBE == 2) ? fbsmi_u2f(bits[k] >> 9) : normal_from_bits(bits[k]);
        const float d = em_drift<MODE>(a.cx, a.cs, g.x[k], g.s[k]);
        const float m = g.x[k] + d * a.dt;
        const float nz = a.sd * z;
        y[k] = m + nz;
    }
    float4 out = make_float4(y[0], y[1], y[2], y[3]);
    if ((int32_t)g.r == a.pin_row) {
        const f4u pv = *(const f4u*)(a.pin_value + g.p);
        out = make_float4(pv.x, pv.y, pv.z, pv.w);
    }
    *(float4*)(a.us_new + (int64_t)g.r * a.du + g.p) = out;
}

// the same for any du and any alignment, element by element
template <int NETDT, int MODE>
__device__ __forceinline__ void em_u_scalar(const EmArgs& a, uint32_t e0, uint32_t lim, const uint32_t (&bits)[4]) {
    const uint32_t le = e0 - a.first_el;
    const uint32_t du = (uint32_t)a.du;
    uint32_t r = le / du, p = le - r * du;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e0 + k < lim) {
            const int64_t src = a.A ? (int64_t)a.A[r] : (int64_t)r;
            const float x = a.us[src * du + p];
            const float s = net_at<NETDT>(a.net, (int64_t)(a.net_A ? a.net_A[r] : (int32_t)r) * a.D + a.u_off[p]);
            const float z = normal_from_bits(bits[k]);
            const float d = em_drift<MODE>(a.cx, a.cs, x, s);
            const float m = x + d * a.dt;
            const float nz = a.sd * z;
            float y = m + nz;
            if ((int32_t)r == a.pin_row) y = a.pin_value[p];
            a.us_new[(int64_t)r * du + p] = y;
        }
        if (++p == du) { p = 0; ++r; }
    }
}

// ------------------------------------------------------------------------------------------------
// log-density of one row: sum over the d elements of a part (PART 0: observed pixels, base v_prev[j]
// shared by all rows; PART 1: unobserved pixels, base us[row][j]) of
//     ( log(2 pi sd^2) + (target[j] - (base + drift * dt))^2 / sd^2 ) / -2
// in the pairwise order of orc_sum (adjacent pairs, zero padded).  `seg` = LDS scratch of
// max(64, segments) floats.  Every thread of the workgroup must call it; thread 0 returns the sum.
// ------------------------------------------------------------------------------------------------
// One 256-element segment of a row's log-density in flight: its offset words, then its operands.
struct SegTab {
    int4 o4;
    int j0, jl;  // first element of the lane's group; the same clamped into the row (loads are never predicated)
    bool full;   // the lane's four elements exist (false for every lane of a segment past the row's end)
};
struct SegOps {
    float s[4];
    f4u tg, b4;
    int j0;
    bool full;
};

template <int NETDT, int MODE, int PART>
__device__ __forceinline__ float em_row_logpdf(const EmArgs& a, int32_t r, float* seg) {
    const int d = PART == 0 ? a.dv : a.du;
    const int32_t* __restrict__ offt = PART == 0 ? a.v_off : a.u_off;
    const float* __restrict__ basep = PART == 0 ? a.v_prev : a.us + (int64_t)r * a.du;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = (d + 255) >> 8;
    const float var = a.sd * a.sd;
    const float rvar = 1.0f / var;  // for div_by(): every quotient of this row has the divisor var
    const float lognorm = fbsmi_logf(6.2831855f * var);
    const int64_t row_base = (int64_t)(a.net_A ? a.net_A[r] : r) * a.D;
    const int dl = d >= 4 ? ((d - 4) & ~3) : 0;   // last whole group of the row (the offset table has >= 4 entries: see launch)

    auto table = [&](int sg) {      // stage 1: the segment's offset words
        SegTab t;
        t.j0 = sg * 256 + lane * 4;
        t.full = sg < nseg && t.j0 + 3 < d;
        t.jl = t.j0 < dl ? t.j0 : dl;
        t.o4 = *(const int4*)(offt + t.jl);
        return t;
    };
    auto issue = [&](const SegTab& t) {  // stage 2: network output (one 16-byte load inside a run of the mask, four loads
        SegOps o;                        // for a group across two runs), target and base; nothing waits here
        o.j0 = t.j0;
        o.full = t.full;
        const int of[4] = {t.o4.x, t.o4.y, t.o4.z, t.o4.w};
        net_at4<NETDT>(a.net, row_base, a.D, of, t.full, o.s);
        o.tg = *(const f4u*)(a.v + t.jl);          // rows of a (T+1, d) path: dword aligned only
        o.b4 = *(const f4u*)(basep + t.jl);
        return o;
    };
    auto compute = [&](const SegOps& o, int sg) {  // stage 3: the arithmetic, the wave's tree, the segment sum
        float t[4];
        if (o.full) {
            const float bb[4] = {o.b4.x, o.b4.y, o.b4.z, o.b4.w};
            const float g[4] = {o.tg.x, o.tg.y, o.tg.z, o.tg.w};
            if (FBSMI_EM_PROBE == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = bb[k] + g[k] + o.s[k];
            } else {
                float sq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dr = em_drift<MODE>(a.cx, a.cs, bb[k], o.s[k]);
                    const float m = bb[k] + dr * a.dt;
                    const float df = g[k] - m;
                    sq[k] = df * df;
                }
                // (df * df) / var, correctly rounded: by the reciprocal when the wave's operands are in its range
                const bool lean = __all(div_by_in_range(fmaxf(fmaxf(sq[0], sq[1]), fmaxf(sq[2], sq[3]))) &&
                                        div_by_in_range(fminf(fminf(sq[0], sq[1]), fminf(sq[2], sq[3]))));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float q = lean ? div_by(sq[k], var, rvar) : sq[k] / var;
                    t[k] = (lognorm + q) * -0.5f;
                }
            }
        } else {                                   // the row's ragged tail: at most one lane per row
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = 0.0f;
                if (o.j0 + k < d) {
                    const float sv = net_at<NETDT>(a.net, row_base + offt[o.j0 + k]);
                    const float bv = basep[o.j0 + k];
                    const float dr = em_drift<MODE>(a.cx, a.cs, bv, sv);
                    const float m = bv + dr * a.dt;
                    const float df = a.v[o.j0 + k] - m;
                    const float q = (df * df) / var;
                    t[k] = (lognorm + q) * -0.5f;
                }
            }
        }
        float sum = (t[0] + t[1]) + (t[2] + t[3]);
        TreePath path;  // the sibling records are not needed here; the compiler drops them
        sum = wave_upsweep(sum, path);
        if (lane == 0) seg[sg] = sum;
    };

    // A wave owns segments wave, wave + 4, ...  The role is bound by memory latency, not by arithmetic (its time did not
    // change when the arithmetic was removed), and a wave's loads retire in issue order, so the three stages are
    // software-pipelined: while segment i is computed the operands of segment i + 1 and the offset words of segment
    // i + 2 are in flight, and a wave never drains its queue before the last segment.
    int sg = wave;
    if (d < 4) {  // fewer elements than one group: the element-wise path only (wave-uniform)
        SegOps none;
        none.j0 = lane * 4;
        none.full = false;
        if (sg < nseg) compute(none, sg);
        sg = nseg;
    }
    SegTab ta, tb;
    SegOps oa, ob;
    if (sg < nseg) {
        ta = table(sg);
        tb = table(sg + kWaves);
        oa = issue(ta);
    }
    while (sg < nseg) {
        ta = table(sg + 2 * kWaves);
        ob = issue(tb);
        compute(oa, sg);
        sg += kWaves;
        if (sg >= nseg) break;
        tb = table(sg + 2 * kWaves);
        oa = issue(ta);
        compute(ob, sg);
        sg += kWaves;
        if (sg >= nseg) break;
    }
    __syncthreads();
    float root = 0.0f;
    if (nseg <= 64) {
        if (wave == 0) {
            float x = lane < nseg ? seg[lane] : 0.0f;
            TreePath path;
            root = wave_upsweep(x, path);
        }
    } else {
        // pairwise levels in LDS, ping-pong between the two halves of `seg` (2 * nseg floats)
        float* cur = seg;
        float* nxt = seg + nseg;
        int m = nseg;
        while (m > 1) {
            const int h = m >> 1;
            for (int i = threadIdx.x; i < h; i += kBlock) nxt[i] = cur[2 * i] + cur[2 * i + 1];
            if ((m & 1) && threadIdx.x == 0) nxt[h] = cur[m - 1];
            m = h + (m & 1);
            __syncthreads();
            float* tsw = cur; cur = nxt; nxt = tsw;
        }
        root = cur[0];
    }
    return root;
}

// Which role workgroup b plays: proposal groups are spread evenly among the log-density groups.
// (vfirst: longest jobs first.  With few workgroups per CU the evenly interleaved order leaves a tail of log-density rows:
// 48.8 -> 46.2 us at config 5's per-GPU shape, 45.9 -> 41.0 with a bfloat16 network; with 8x the rows the interleave wins.)
__device__ __forceinline__ bool em_role(int b, int nU, int total, int& index, int vfirst = 0) {
    if (vfirst) {
        const int nV = total - nU;
        const bool is_u = b >= nV;
        index = is_u ? b - nV : b;
        return is_u;
    }
    const int u0 = (int)(((int64_t)b * nU) / total);
    const int u1 = (int)(((int64_t)(b + 1) * nU) / total);
    const bool is_u = u1 != u0;
    index = is_u ? u0 : b - u0;
    return is_u;
}

// PAIR: the launch covers the whole draw (row0 == 0, n == n_total): a proposal thread owns flat elements
// i..i+3 and i+half..i+half+3, the two words of the same block-cipher calls.
template <bool PAIR, bool VEC, int NETDT, int MODE>
__global__ void __launch_bounds__(kBlock, FBSMI_EM_WAVES) k_em_finish(const EmArgs a) {
    extern __shared__ float seg[];
    int index;
    const bool is_u = em_role(blockIdx.x, a.nU, a.nU + a.nV, index, a.vfirst);
    if (is_u) {
        const uint32_t g = (uint32_t)index * kBlock + threadIdx.x;
        if (PAIR) {
            const uint32_t i0 = 4u * g;
            if (i0 >= a.half) return;
            UGroup g0, g1;
            if (VEC) {  // every load of both groups is issued before the first random word is computed
                em_u_load<NETDT>(a, i0, g0);
                em_u_load<NETDT>(a, i0 + a.half, g1);
            }
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = i0 + k, j = i + a.half;
                if (FBSMI_EM_PROBE == 2) { lo[k] = i; hi[k] = j; }
                else threefry2x32(a.k0, a.k1, i, j < a.ntot_el ? j : 0u, lo[k], hi[k]);
            }
            if (VEC) {
                em_u_store<NETDT, MODE>(a, g0, lo);
                em_u_store<NETDT, MODE>(a, g1, hi);
            } else {
                em_u_scalar<NETDT, MODE>(a, i0, a.half, lo);
                em_u_scalar<NETDT, MODE>(a, i0 + a.half, a.ntot_el, hi);
            }
        } else {
            const uint32_t l0 = 4u * g;
            if (l0 >= a.nloc_el) return;
            const uint32_t e0 = a.first_el + l0;
            UGroup g0;
            if (VEC) em_u_load<NETDT>(a, e0, g0);
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = random_bits_at(a.k0, a.k1, a.ntot_el, (uint64_t)e0 + k);
            if (VEC) em_u_store<NETDT, MODE>(a, g0, w);
            else em_u_scalar<NETDT, MODE>(a, e0, a.first_el + a.nloc_el, w);
        }
    } else {
        const int row = index;
        const float root = em_row_logpdf<NETDT, MODE, 0>(a, row, seg);
        if (threadIdx.x == 0) a.lw[row] = root;
    }
}

// ------------------------------------------------------------------------------------------------
// The same step as ONE pass over whole rows of the network output (k_em_rows).  Measured on this chip: reading the
// observed 75 % of a row through the offset table costs as much DRAM time as reading the whole row, and the two roles
// of k_em_finish do not overlap (both sit on the memory pipeline).  Here a workgroup owns a row: the row of the network
// output goes to LDS by LDS-DMA (global_load_lds_dwordx4: full-width coalesced, no registers, all of it in flight at
// once) while the workgroup computes its share of the row's normals; after one barrier both parts of the row are served
// from LDS through the offset tables -- proposal (4 elements per thread and group of 1024), then the log-density
// segments (pipelined as in em_row_logpdf, the table / target / base from L2).  PAIR2: the launch covers the whole draw
// with an even number of rows; the workgroup then takes rows r and r + n/2 one after the other and keeps the second
// words of its Threefry calls in registers for the second row (the pairing of jax's random_bits).
// Requirements (else k_em_finish): du % 4 == 0, du <= 4096, row bytes a multiple of 16 that fit the LDS budget.
// ------------------------------------------------------------------------------------------------
template <int NETDT>
__device__ __forceinline__ float lds_net(const void* lnet, int off) {
    if (NETDT == 0) return ((const float*)lnet)[off];
    return bf16_to_f32(((const unsigned short*)lnet)[off]);
}

// four staged values; one wide LDS read when the lane's offsets are consecutive and aligned (conflict-free: neighbouring
// lanes read neighbouring 16-byte slots), else four 4-way conflicting dword reads
template <int NETDT>
__device__ __forceinline__ void lds_net4(const void* lnet, const int (&of)[4], float (&s)[4]) {
    if (of[3] - of[0] == 3 && (of[0] & 3) == 0) {
        if (NETDT == 0) {
            const float4 t = *(const float4*)((const float*)lnet + of[0]);
            s[0] = t.x; s[1] = t.y; s[2] = t.z; s[3] = t.w;
        } else {
            const ushort4 t = *(const ushort4*)((const unsigned short*)lnet + of[0]);
            s[0] = bf16_to_f32(t.x); s[1] = bf16_to_f32(t.y); s[2] = bf16_to_f32(t.z); s[3] = bf16_to_f32(t.w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = lds_net<NETDT>(lnet, of[k]);
    }
}

// stage the row of `net` that belongs to local row r; the caller waits (vmcnt(0) + barrier) before reading it
template <int NETDT>
__device__ __forceinline__ void em_stage_row(const EmArgs& a, int32_t r, void* lnet) {
    const int esz = NETDT == 0 ? 4 : 2;
    const int64_t row = (int64_t)(a.net_A ? a.net_A[r] : r) * a.D;
    const char* g = (const char*)a.net + row * esz;
    const int rowbytes = a.D * esz;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c * 1024 < rowbytes; c += kWaves) {
        const int b = c * 1024 + lane * 16;
        if (b < rowbytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + b),
                                             (__attribute__((address_space(3))) void*)((char*)lnet + c * 1024), 16, 0, 0);
    }
}

// log-density of the staged row (PART 0 of em_row_logpdf with the network values read from LDS)
template <int NETDT, int MODE>
__device__ __forceinline__ float em_row_logpdf_lds(const EmArgs& a, const void* lnet, float* seg) {
    const int d = a.dv;
    const int32_t* __restrict__ offt = a.v_off;
    const float* __restrict__ basep = a.v_prev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = (d + 255) >> 8;
    const float var = a.sd * a.sd;
    const float rvar = 1.0f / var;
    const float lognorm = fbsmi_logf(6.2831855f * var);
    const int dl = d >= 4 ? ((d - 4) & ~3) : 0;
    struct Ops { int4 o4; f4u tg, b4; int j0; bool full; };
    auto issue = [&](int sg) {
        Ops o;
        o.j0 = sg * 256 + lane * 4;
        o.full = sg < nseg && o.j0 + 3 < d;
        const int jl = o.j0 < dl ? o.j0 : dl;
        o.o4 = *(const int4*)(offt + jl);
        o.tg = *(const f4u*)(a.v + jl);
        o.b4 = *(const f4u*)(basep + jl);
        return o;
    };
    auto compute = [&](const Ops& o, int sg) {
        float t[4];
        if (o.full) {
            const int of[4] = {o.o4.x, o.o4.y, o.o4.z, o.o4.w};
            const float bb[4] = {o.b4.x, o.b4.y, o.b4.z, o.b4.w};
            const float g[4] = {o.tg.x, o.tg.y, o.tg.z, o.tg.w};
            float sq[4], sv4[4];
            lds_net4<NETDT>(lnet, of, sv4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sv = sv4[k];
                const float dr = em_drift<MODE>(a.cx, a.cs, bb[k], sv);
                const float m = bb[k] + dr * a.dt;
                const float df = g[k] - m;
                sq[k] = df * df;
            }
            const bool lean = __all(div_by_in_range(fmaxf(fmaxf(sq[0], sq[1]), fmaxf(sq[2], sq[3]))) &&
                                    div_by_in_range(fminf(fminf(sq[0], sq[1]), fminf(sq[2], sq[3]))));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float q = lean ? div_by(sq[k], var, rvar) : sq[k] / var;
                t[k] = (lognorm + q) * -0.5f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = 0.0f;
                if (o.j0 + k < d) {
                    const float sv = lds_net<NETDT>(lnet, offt[o.j0 + k]);
                    const float bv = basep[o.j0 + k];
                    const float dr = em_drift<MODE>(a.cx, a.cs, bv, sv);
                    const float m = bv + dr * a.dt;
                    const float df = a.v[o.j0 + k] - m;
                    const float q = (df * df) / var;
                    t[k] = (lognorm + q) * -0.5f;
                }
            }
        }
        float sum = (t[0] + t[1]) + (t[2] + t[3]);
        TreePath path;
        sum = wave_upsweep(sum, path);
        if (lane == 0) seg[sg] = sum;
    };
    int sg = wave;
    if (d < 4) {
        Ops none;
        none.j0 = lane * 4;
        none.full = false;
        if (sg < nseg) compute(none, sg);
        sg = nseg;
    }
    Ops oa, ob;
    if (sg < nseg) oa = issue(sg);
    while (sg < nseg) {
        ob = issue(sg + kWaves);
        compute(oa, sg);
        sg += kWaves;
        if (sg >= nseg) break;
        oa = issue(sg + kWaves);
        compute(ob, sg);
        sg += kWaves;
    }
    __syncthreads();
    float root = 0.0f;
    if (nseg <= 64) {
        if (wave == 0) {
            float x = lane < nseg ? seg[lane] : 0.0f;
            TreePath path;
            root = wave_upsweep(x, path);
        }
    } else {
        float* cur = seg;
        float* nxt = seg + nseg;
        int m = nseg;
        while (m > 1) {
            const int h = m >> 1;
            for (int i = threadIdx.x; i < h; i += kBlock) nxt[i] = cur[2 * i] + cur[2 * i + 1];
            if ((m & 1) && threadIdx.x == 0) nxt[h] = cur[m - 1];
            m = h + (m & 1);
            __syncthreads();
            float* tsw = cur; cur = nxt; nxt = tsw;
        }
        root = cur[0];
    }
    return root;
}

template <int KU, bool PAIR2, int NETDT, int MODE>
__global__ void __launch_bounds__(kBlock, 3) k_em_rows(const EmArgs a, int lnet_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    void* lnet = (void*)lds_raw;
    float* seg = (float*)(lds_raw + lnet_bytes);
    const uint32_t du = (uint32_t)a.du;
    const int32_t r0 = blockIdx.x;                      // first (or only) row of this workgroup
    const int nrows = PAIR2 ? 2 : 1;
    const int32_t half_rows = a.n >> 1;

    em_stage_row<NETDT>(a, r0, lnet);
    // this thread's proposal groups: p = 4 * (tid + 256 k)
    int4 o4[KU];
    uint32_t hi[KU][4];
    float z[KU][4];
    float4 x4[KU];
    {
        const int64_t src = a.A ? (int64_t)a.A[r0] : (int64_t)r0;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const uint32_t p = 4u * (threadIdx.x + 256u * k);
            const uint32_t pl = p < du ? p : du - 4;    // clamped: loads are never predicated
            o4[k] = *(const int4*)(a.u_off + pl);
            x4[k] = *(const float4*)(a.us + src * du + pl);
        }
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const uint32_t p = 4u * (threadIdx.x + 256u * k);
            const uint32_t e0 = a.first_el + (uint32_t)r0 * du + p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t lo;
                if (FBSMI_EM_PROBE == 2) { lo = e0 + j; hi[k][j] = lo; }
                else if (PAIR2) threefry2x32(a.k0, a.k1, e0 + j, e0 + j + a.half, lo, hi[k][j]);
                else lo = random_bits_at(a.k0, a.k1, a.ntot_el, (uint64_t)e0 + j);
                z[k][j] = (FBSMI_EM_PROBE == 1 || FBSMI_EM_PROBE == 2) ? fbsmi_u2f(lo >> 9) : normal_from_bits(lo);
            }
        }
    }
    for (int rr = 0; rr < nrows; ++rr) {
        const int32_t r = rr == 0 ? r0 : r0 + half_rows;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of this row (and the operand loads) have landed
        __syncthreads();
        // ---- proposal for row r ----
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            const uint32_t p = 4u * (threadIdx.x + 256u * k);
            if (p < du) {
                const int of[4] = {o4[k].x, o4[k].y, o4[k].z, o4[k].w};
                const float xs[4] = {x4[k].x, x4[k].y, x4[k].z, x4[k].w};
                float y[4], sv4[4];
                lds_net4<NETDT>(lnet, of, sv4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sv = sv4[j];
                    const float dr = em_drift<MODE>(a.cx, a.cs, xs[j], sv);
                    const float m = xs[j] + dr * a.dt;
                    const float nz = a.sd * z[k][j];
                    y[j] = m + nz;
                }
                float4 out = make_float4(y[0], y[1], y[2], y[3]);
                if (r == a.pin_row) {
                    const f4u pv = *(const f4u*)(a.pin_value + p);
                    out = make_float4(pv.x, pv.y, pv.z, pv.w);
                }
                *(float4*)(a.us_new + (int64_t)r * du + p) = out;
            }
        }
        // ---- log-density of row r ----
        const float root = FBSMI_EM_PROBE == 5 ? 0.0f : em_row_logpdf_lds<NETDT, MODE>(a, lnet, seg);
        if (threadIdx.x == 0) a.lw[r] = root;
        if (PAIR2 && rr == 0) {
            __syncthreads();                                 // everyone is done with the staged row and the segment sums
            const int32_t r2 = r0 + half_rows;
            em_stage_row<NETDT>(a, r2, lnet);
            const int64_t src = a.A ? (int64_t)a.A[r2] : (int64_t)r2;
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const uint32_t p = 4u * (threadIdx.x + 256u * k);
                const uint32_t pl = p < du ? p : du - 4;
                x4[k] = *(const float4*)(a.us + src * du + pl);
            }
#pragma unroll
            for (int k = 0; k < KU; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    z[k][j] = (FBSMI_EM_PROBE == 1 || FBSMI_EM_PROBE == 2) ? fbsmi_u2f(hi[k][j] >> 9) : normal_from_bits(hi[k][j]);
        }
    }
}

template <int NETDT, int MODE>
__global__ void __launch_bounds__(kBlock) k_em_translp(const EmArgs a) {
    extern __shared__ float seg[];
    const float root = em_row_logpdf<NETDT, MODE, 1>(a, blockIdx.x, seg);
    if (threadIdx.x == 0) a.lw[blockIdx.x] = root;
}

// img[r][e] = role[e] >= 0 ? us[A[r]][role[e]] : v_prev[~role[e]].  One workgroup = one row x kCatGroups chunks of
// 1024 elements; a thread fetches the role words of all its chunks, then every source element, then stores: a few
// KB in flight per wave, because a workgroup that moved 16 bytes per thread spent its life waiting (84 % of its wave
// cycles, rocprofv3 SQ_WAIT_ANY) on two dependent round trips.
constexpr int kCatGroups = 4;

template <int OUTDT, bool VEC>
__global__ void __launch_bounds__(kBlock) k_em_concat(const float* __restrict__ us, const int32_t* __restrict__ A,
                                                     const float* __restrict__ v_prev, const int32_t* __restrict__ role,
                                                     int32_t du, int32_t D, int32_t chunks, void* img) {
    const int32_t r = blockIdx.x / chunks, c = blockIdx.x - r * chunks;
    const int base = c * (1024 * kCatGroups) + threadIdx.x * 4;
    const float* __restrict__ row = us + (int64_t)(A ? A[r] : r) * du;
    if (VEC) {
        // role words of all groups, then one 16-byte load per group wherever its four sources are consecutive (inside a
        // run of the mask: four unobserved elements of the ancestor's row, or four observed ones of v_prev) -- the address
        // unit takes four lanes per cycle whatever the width, so element-wise gathers cost four times as much; groups
        // across two runs take the element-wise path, for the whole wave (loads are never predicated per lane)
        int4 ro[kCatGroups];
        const int dl = (D - 4) & ~3;
#pragma unroll
        for (int g = 0; g < kCatGroups; ++g) {
            const int e = base + g * 1024;
            ro[g] = *(const int4*)(role + (e < dl ? e : dl));
        }
        float x[kCatGroups][4];
#pragma unroll
        for (int g = 0; g < kCatGroups; ++g) {
            const bool valid = base + g * 1024 < D;
            const int o0 = ro[g].x, o3 = ro[g].w;
            const bool urun = o0 >= 0 && o3 == o0 + 3, vrun = o0 < 0 && o3 == o0 - 3;
            // clamped so that the wide load stays inside its array whatever the lane holds
            const int uo = o0 >= 0 ? (o0 < du - 4 ? o0 : du - 4) : 0;
            const int vq = o0 < 0 ? ((~o0) < (D - du) - 4 ? (~o0) : (D - du) - 4) : 0;
            const float* __restrict__ src = o0 >= 0 ? row + uo : v_prev + (vq > 0 ? vq : 0);
            const f4u w = *(const f4u*)src;
            x[g][0] = w.x; x[g][1] = w.y; x[g][2] = w.z; x[g][3] = w.w;
            if (__any(valid && !(urun || vrun))) {
                const int o[4] = {ro[g].x, ro[g].y, ro[g].z, ro[g].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* __restrict__ s1 = o[k] >= 0 ? row + o[k] : v_prev + ~o[k];
                    const float xv = *s1;
                    x[g][k] = (urun || vrun) ? x[g][k] : xv;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < kCatGroups; ++g)
            if (base + g * 1024 < D) {
                const int64_t at = (int64_t)r * D + base + g * 1024;
                if (OUTDT == 0) {
                    *(float4*)((float*)img + at) = make_float4(x[g][0], x[g][1], x[g][2], x[g][3]);
                } else {
                    ushort4 h;
                    h.x = f32_to_bf16(x[g][0]); h.y = f32_to_bf16(x[g][1]);
                    h.z = f32_to_bf16(x[g][2]); h.w = f32_to_bf16(x[g][3]);
                    *(ushort4*)((unsigned short*)img + at) = h;
                }
            }
    } else {
#pragma unroll
        for (int g = 0; g < kCatGroups; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = base + g * 1024 + k;
                if (e < D) {
                    const int o = role[e];
                    const float xv = o >= 0 ? row[o] : v_prev[~o];
                    if (OUTDT == 0) ((float*)img)[(int64_t)r * D + e] = xv;
                    else ((unsigned short*)img)[(int64_t)r * D + e] = f32_to_bf16(xv);
                }
            }
    }
}

#define FBSMI_NEED(cond, msg) \
    if (!(cond)) return fail(FBSMI_ERR_ARG, msg)
#define FBSMI_LAUNCH_CHECK()                                                     \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

template <bool PAIR, bool VEC>
int launch_finish(const EmArgs& a, int net_dtype, int mode, size_t lds, hipStream_t st) {
    const dim3 grid((unsigned)(a.nU + a.nV));
    if (net_dtype == 0 && mode == 0) k_em_finish<PAIR, VEC, 0, 0><<<grid, kBlock, lds, st>>>(a);
    else if (net_dtype == 0) k_em_finish<PAIR, VEC, 0, 1><<<grid, kBlock, lds, st>>>(a);
    else if (mode == 0) k_em_finish<PAIR, VEC, 1, 0><<<grid, kBlock, lds, st>>>(a);
    else k_em_finish<PAIR, VEC, 1, 1><<<grid, kBlock, lds, st>>>(a);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

template <int KU, bool PAIR2>
int launch_rows(const EmArgs& a, int net_dtype, int mode, int lnet_bytes, size_t lds, hipStream_t st) {
    const dim3 grid((unsigned)(PAIR2 ? a.n / 2 : a.n));
    if (net_dtype == 0 && mode == 0) k_em_rows<KU, PAIR2, 0, 0><<<grid, kBlock, lds, st>>>(a, lnet_bytes);
    else if (net_dtype == 0) k_em_rows<KU, PAIR2, 0, 1><<<grid, kBlock, lds, st>>>(a, lnet_bytes);
    else if (mode == 0) k_em_rows<KU, PAIR2, 1, 0><<<grid, kBlock, lds, st>>>(a, lnet_bytes);
    else k_em_rows<KU, PAIR2, 1, 1><<<grid, kBlock, lds, st>>>(a, lnet_bytes);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

template <bool PAIR2>
int launch_rows_ku(int ku, const EmArgs& a, int net_dtype, int mode, int lnet_bytes, size_t lds, hipStream_t st) {
    switch (ku) {
        case 1: return launch_rows<1, PAIR2>(a, net_dtype, mode, lnet_bytes, lds, st);
        case 2: return launch_rows<2, PAIR2>(a, net_dtype, mode, lnet_bytes, lds, st);
        case 3: return launch_rows<3, PAIR2>(a, net_dtype, mode, lnet_bytes, lds, st);
        default: return launch_rows<4, PAIR2>(a, net_dtype, mode, lnet_bytes, lds, st);
    }
}

size_t seg_lds_bytes(int d) {
    const int nseg = (d + 255) >> 8;
    return sizeof(float) * (size_t)(nseg <= 64 ? 64 : 2 * nseg);
}

bool mask_ok(const fbsmi_em_mask* m) {
    return m && m->du >= 1 && m->dv >= 0 && m->u_off && (m->dv == 0 || m->v_off) && m->role;
}

}  // namespace

}  // namespace fbsmi

using namespace fbsmi;

extern "C" {

int fbsmi_em_concat(const fbsmi_em_mask* mask, const float* us, const int32_t* A, const float* v_prev, int64_t n,
                    int out_dtype, void* img, void* stream) {
    FBSMI_NEED(mask_ok(mask) && n >= 0 && (out_dtype == 0 || out_dtype == 1), "em_concat: bad arguments");
    if (n == 0) return FBSMI_OK;
    FBSMI_NEED(us && img && (mask->dv == 0 || v_prev), "em_concat: null pointer");
    const int32_t D = mask->du + mask->dv;
    const int32_t chunks = (D + 1024 * kCatGroups - 1) / (1024 * kCatGroups);
    FBSMI_NEED(n * chunks < (int64_t)1 << 31, "em_concat: too many rows");
    const bool vec = D % 4 == 0 && mask->du >= 4 && mask->dv >= 4 && ((uintptr_t)img & 15) == 0 &&
                     ((uintptr_t)mask->role & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(n * chunks));
#define FBSMI_CC(OD, V) \
    k_em_concat<OD, V><<<grid, kBlock, 0, st>>>(us, A, v_prev, mask->role, mask->du, D, chunks, img)
    if (out_dtype == 0) { if (vec) FBSMI_CC(0, true); else FBSMI_CC(0, false); }
    else { if (vec) FBSMI_CC(1, true); else FBSMI_CC(1, false); }
#undef FBSMI_CC
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_em_finish(const fbsmi_em_mask* mask, const float* us, const int32_t* A, const void* net,
                    const int32_t* net_A, int net_dtype, int mode, float cx, float cs, float dt, float sd, const float* v, const float* v_prev, uint32_t k0,
                    uint32_t k1, int64_t n_total, int64_t row0, int64_t n, int64_t pin_row, const float* pin_value,
                    float* us_new, float* lw, void* stream) {
    FBSMI_NEED(mask_ok(mask) && n >= 0 && row0 >= 0 && row0 + n <= n_total && (net_dtype == 0 || net_dtype == 1) &&
                   (mode == 0 || mode == 1), "em_finish: bad arguments");
    if (n == 0 || (!us_new && !lw)) return FBSMI_OK;
    FBSMI_NEED(net && n < ((int64_t)1 << 31), "em_finish: null network output or too many rows");
    FBSMI_NEED(mask->du + mask->dv >= 4, "em_finish: images of fewer than 4 floats are not supported");
    FBSMI_NEED(!us_new || (us && us_new != us), "em_finish: the proposal needs us, and us_new must not alias it");
    FBSMI_NEED(!lw || mask->dv == 0 || (v && v_prev), "em_finish: the weights need v and v_prev");
    FBSMI_NEED(pin_row < 0 || (pin_row < n && pin_value), "em_finish: bad pin");
    FBSMI_NEED(n_total * (int64_t)mask->du < ((int64_t)1 << 32), "em_finish: the noise draw exceeds 2^32 elements");
    EmArgs a;
    a.us = us; a.A = A; a.net_A = net_A; a.net = net; a.v = v; a.v_prev = v_prev; a.pin_value = pin_value;
    a.us_new = us_new; a.lw = lw; a.u_off = mask->u_off; a.v_off = mask->v_off;
    a.du = mask->du; a.dv = mask->dv; a.D = mask->du + mask->dv;
    a.cx = cx; a.cs = cs; a.dt = dt; a.sd = sd; a.k0 = k0; a.k1 = k1;
    a.ntot_el = (uint32_t)(n_total * mask->du);
    a.half = (uint32_t)(((uint64_t)a.ntot_el + 1) >> 1);
    a.first_el = (uint32_t)(row0 * mask->du);
    a.nloc_el = (uint32_t)(n * mask->du);
    a.n = (int32_t)n;
    a.pin_row = us_new && pin_row >= 0 ? (int32_t)pin_row : -1;
    const bool pair = row0 == 0 && n == n_total;
    const uint32_t groups = pair ? (a.half + 3) / 4 : (a.nloc_el + 3) / 4;
    a.nU = us_new ? (int32_t)((groups + kBlock - 1) / kBlock) : 0;
    a.nV = lw ? (int32_t)n : 0;
    a.vfirst = (a.nU > 0 && a.nV > 0 && a.nU + a.nV <= 8192) ? 1 : 0;   // up to ~4 workgroups per CU slot: rows first
    // 16-byte accesses on particle rows: whole rows of 4-float groups, aligned bases, and in the paired
    // walk a second half that starts on a group boundary
    const bool vec = mask->du % 4 == 0 && (!pair || a.half % 4 == 0) && (a.first_el % 4 == 0) &&
                     ((((uintptr_t)us | (uintptr_t)us_new) & 15) == 0);
    FBSMI_NEED((((uintptr_t)mask->u_off | (uintptr_t)mask->v_off) & 15) == 0,
               "em_finish: the mask tables must be 16-byte aligned");
    const size_t lds = seg_lds_bytes(mask->dv);
    hipStream_t st = (hipStream_t)stream;
    // A rank's row slice of a sharded ensemble cannot pair the Threefry words, and its proposal groups then cost the
    // two-role kernel 96 us at the config-5 share; one pass over whole rows (k_em_rows) does it in 56.  The whole draw
    // stays on the two-role kernel (49 us against 62).  FBSMI_EM_ROWS=0 forces the two-role kernel (diagnostics).
    {
        static const int rows_on = [] { const char* e = getenv("FBSMI_EM_ROWS"); return e ? atoi(e) : 1; }();
        const int esz = net_dtype == 0 ? 4 : 2;
        const int64_t rowbytes = (int64_t)a.D * esz;
        const int lnet_bytes = (int)((rowbytes + 1023) / 1024 * 1024);
        const bool rows_ok = rows_on && !pair && us_new && lw && vec && mask->dv >= 1 && mask->du <= 4096 && rowbytes % 16 == 0 &&
                             lnet_bytes + (int64_t)lds <= 52 * 1024 && (((uintptr_t)net) & 15) == 0 && n >= 64;
        if (rows_ok) {
            const int ku = (mask->du + 1023) / 1024;
            const size_t tot = (size_t)lnet_bytes + lds;
            return launch_rows_ku<false>(ku, a, net_dtype, mode, lnet_bytes, tot, st);
        }
    }
    if (pair) return vec ? launch_finish<true, true>(a, net_dtype, mode, lds, st)
                         : launch_finish<true, false>(a, net_dtype, mode, lds, st);
    return vec ? launch_finish<false, true>(a, net_dtype, mode, lds, st)
               : launch_finish<false, false>(a, net_dtype, mode, lds, st);
}

int fbsmi_em_transition_logpdf(const fbsmi_em_mask* mask, const float* us, const void* net, int net_dtype, int mode,
                               float cx, float cs, float dt, float sd, const float* u, int64_t n, float* lw,
                               void* stream) {
    FBSMI_NEED(mask_ok(mask) && n >= 0 && (net_dtype == 0 || net_dtype == 1) && (mode == 0 || mode == 1),
               "em_transition_logpdf: bad arguments");
    if (n == 0) return FBSMI_OK;
    FBSMI_NEED(us && net && u && lw && n < ((int64_t)1 << 31), "em_transition_logpdf: null pointer");
    FBSMI_NEED(mask->du + mask->dv >= 4, "em_transition_logpdf: images of fewer than 4 floats are not supported");
    FBSMI_NEED(((uintptr_t)mask->u_off & 15) == 0, "em_transition_logpdf: the mask tables must be 16-byte aligned");
    EmArgs a = {};
    a.us = us; a.net = net; a.v = u; a.lw = lw; a.u_off = mask->u_off; a.v_off = mask->v_off;
    a.du = mask->du; a.dv = mask->dv; a.D = mask->du + mask->dv;
    a.cx = cx; a.cs = cs; a.dt = dt; a.sd = sd; a.n = (int32_t)n; a.pin_row = -1;
    const size_t lds = seg_lds_bytes(mask->du);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n);
    if (net_dtype == 0 && mode == 0) k_em_translp<0, 0><<<grid, kBlock, lds, st>>>(a);
    else if (net_dtype == 0) k_em_translp<0, 1><<<grid, kBlock, lds, st>>>(a);
    else if (mode == 0) k_em_translp<1, 0><<<grid, kBlock, lds, st>>>(a);
    else k_em_translp<1, 1><<<grid, kBlock, lds, st>>>(a);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

}  // extern "C"
