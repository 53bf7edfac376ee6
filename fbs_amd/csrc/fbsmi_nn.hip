// fbsmi_nn.hip -- kernels for the score network (fbs_amd/unet.py, the torch restatement of fbs/nn/unet.py).
//
// k_linear_attention: the LinearAttention core (fbs/nn/unet.py:209-245).  In eager torch it is ~10 passes over
// (B, n, heads, 32) tensors (two softmaxes over different axes, two scalings, two einsums, reshapes) and was
// 40 % of the UNet's time; here one workgroup per (image, head) reads q, k, v once each: an online
// max / sum over the tokens for softmax(k), the 32 x 32 context in registers (token chunks staged in
// LDS), then softmax(q) and the context product per token chunk.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fbsmi.h"
#include "../../include/fbsmi_nn.h"
#include "fbsmi_host.h"

namespace fbsmi {

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const __hip_bfloat16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

constexpr int kHd = 32;      // dim_head
constexpr int kChunk = 64;   // tokens per LDS chunk

template <typename T>
__global__ void __launch_bounds__(256) k_linear_attention(const T* __restrict__ qkv, T* __restrict__ out, int n,
                                                          int heads) {
    const int h = blockIdx.x, t = threadIdx.x;
    const int HD = heads * kHd, C3 = 3 * HD;
    const T* base = qkv + (size_t)blockIdx.y * n * C3 + h * kHd;   // + token * C3 + which * HD + d
    T* obase = out + (size_t)blockIdx.y * n * HD + h * kHd;
    __shared__ float red[8][kHd][2];
    __shared__ float mS[kHd], zS[kHd];
    __shared__ float ekS[kChunk][kHd + 1], vS[kChunk][kHd + 1], qS[kChunk][kHd + 1];
    __shared__ float ctx[kHd][kHd + 1];
    // ---- softmax(k) over the tokens: running max and sum per embedding coordinate
    {
        const int dd = t & 31, g = t >> 5;
        float m = -__builtin_inff(), z = 0.0f;
        for (int nn = g; nn < n; nn += 8) {
            const float kv = ldf(base + (size_t)nn * C3 + HD + dd);
            const float mn = fmaxf(m, kv);
            z = z * expf(m - mn) + expf(kv - mn);
            m = mn;
        }
        red[g][dd][0] = m;
        red[g][dd][1] = z;
        __syncthreads();
        if (t < kHd) {
            float M = red[0][t][0];
            for (int g2 = 1; g2 < 8; ++g2) M = fmaxf(M, red[g2][t][0]);
            float Z = 0.0f;
            for (int g2 = 0; g2 < 8; ++g2) Z += red[g2][t][1] * expf(red[g2][t][0] - M);
            mS[t] = M;
            zS[t] = Z;
        }
        __syncthreads();
    }
    // ---- context[d][e] = sum_n softmax(k)[n][d] * v[n][e] / n : thread = 2 d's x 2 e's, token chunks in LDS
    {
        const int d2 = (t >> 4) * 2, e2 = (t & 15) * 2;
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
        for (int c0 = 0; c0 < n; c0 += kChunk) {
            for (int i = t; i < kChunk * kHd; i += 256) {
                const int nl = i >> 5, dd = i & 31, nn = c0 + nl;
                float ek = 0.0f, vv = 0.0f;
                if (nn < n) {
                    ek = expf(ldf(base + (size_t)nn * C3 + HD + dd) - mS[dd]);
                    vv = ldf(base + (size_t)nn * C3 + 2 * HD + dd);
                }
                ekS[nl][dd] = ek;
                vS[nl][dd] = vv;
            }
            __syncthreads();
#pragma unroll 8
            for (int nl = 0; nl < kChunk; ++nl) {
                const float k0 = ekS[nl][d2], k1 = ekS[nl][d2 + 1], v0 = vS[nl][e2], v1 = vS[nl][e2 + 1];
                a00 = fmaf(k0, v0, a00);
                a01 = fmaf(k0, v1, a01);
                a10 = fmaf(k1, v0, a10);
                a11 = fmaf(k1, v1, a11);
            }
            __syncthreads();
        }
        const float s0 = 1.0f / (zS[d2] * (float)n), s1 = 1.0f / (zS[d2 + 1] * (float)n);
        ctx[d2][e2] = a00 * s0;
        ctx[d2][e2 + 1] = a01 * s0;
        ctx[d2 + 1][e2] = a10 * s1;
        ctx[d2 + 1][e2 + 1] = a11 * s1;
        __syncthreads();
    }
    // ---- per token: q = softmax(q) / sqrt(32); out[e] = sum_d context[d][e] q[d]
    {
        const int nl = t >> 2, part = t & 3;
        const float rs = 0.17677669529663687f;   // 1 / sqrt(32)
        for (int c0 = 0; c0 < n; c0 += kChunk) {
            const int nn = c0 + nl;
            float q[8];
            float m = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                q[i] = nn < n ? ldf(base + (size_t)nn * C3 + part * 8 + i) : 0.0f;
                m = fmaxf(m, q[i]);
            }
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                q[i] = expf(q[i] - m);
                sum += q[i];
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            const float sc = rs / sum;
#pragma unroll
            for (int i = 0; i < 8; ++i) qS[nl][part * 8 + i] = q[i] * sc;
            __syncthreads();
            float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int dd = 0; dd < kHd; ++dd) {
                const float qd = qS[nl][dd];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(ctx[dd][part * 8 + j], qd, o[j]);
            }
            if (nn < n) {
#pragma unroll
                for (int j = 0; j < 8; ++j) stf(obase + (size_t)nn * HD + part * 8 + j, o[j]);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// to_qkv (1x1 convolution, no bias) + the LinearAttention core in one kernel, bfloat16 on the matrix cores: the
// (B, n, 3 * heads * 32) tensor between the two never exists.  One workgroup per image, one wave per head.  Per block of 32
// tokens a wave forms K, V (or Q^T) as 32 x 32 accumulator tiles of v_mfma_f32_32x32x16_bf16 straight from the
// token-major activations (a lane's operand fragment is 16 contiguous bytes of a token's row; the weight fragments stay
// in registers), and every later product sums over the ROW index of such a tile, so the tile's registers ARE the next
// product's operand fragment (converted pairwise to bfloat16) -- no LDS, no lane movement:
//   pass 1  K = X Wk^T                       -> m_d = max over tokens (d is the lane's column)
//   pass 2  E = exp(K - m), V = X Wv^T       -> ctx[d][e] += E^T V, Z_d += column sums of E
//   pass 3  Q^T = Wq X^T, softmax over d (the tile's rows: in-lane + one exchange), out^T = ctx'^T Q'^T, 8-byte stores
// with ctx' = ctx / (Z_d n) and Q' = softmax(Q) / sqrt(32) (fbs/nn/unet.py:209-245).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned p;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(lo), "v"(hi));
    return p;
}
template <int S>
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x) {   // registers 8S .. 8S+7 as the fragment of k-step S
    const uint4 u = make_uint4(pk_bf16(x[8 * S], x[8 * S + 1]), pk_bf16(x[8 * S + 2], x[8 * S + 3]),
                               pk_bf16(x[8 * S + 4], x[8 * S + 5]), pk_bf16(x[8 * S + 6], x[8 * S + 7]));
    return __builtin_bit_cast(bf16x8, u);
}

template <int CK>   // C = 16 * CK input channels
__global__ void __launch_bounds__(256) k_qkv_linear_attention(const __hip_bfloat16* __restrict__ xn,
                                                              const __hip_bfloat16* __restrict__ w,
                                                              __hip_bfloat16* __restrict__ out, int n, int heads) {
    constexpr int C = 16 * CK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int HD = heads * kHd;
    const __hip_bfloat16* xb = xn + (size_t)blockIdx.x * n * C;
    __hip_bfloat16* ob = out + (size_t)blockIdx.x * n * HD;
    __shared__ float zs[4][kHd];
    auto load_x = [&](int t0, bf16x8 (&xa)[CK]) {   // operand fragments of tokens t0 .. t0+31 (rows past n: zeros)
        const int tok = t0 + r;
        const bool ok = tok < n;
        const uint4* p = reinterpret_cast<const uint4*>(xb + (size_t)(ok ? tok : n - 1) * C + 8 * h2);
#pragma unroll
        for (int s = 0; s < CK; ++s) {
            uint4 u = p[2 * s];
            if (!ok) u = make_uint4(0u, 0u, 0u, 0u);
            xa[s] = __builtin_bit_cast(bf16x8, u);
        }
    };
    auto load_w = [&](int which, int head, bf16x8 (&wf)[CK]) {
        const uint4* p = reinterpret_cast<const uint4*>(w + (size_t)(which * HD + head * kHd + r) * C + 8 * h2);
#pragma unroll
        for (int s = 0; s < CK; ++s) wf[s] = __builtin_bit_cast(bf16x8, p[2 * s]);
    };
    auto row_of = [&](int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * h2; };
    const int iters = (heads + 3) >> 2;
    for (int it = 0; it < iters; ++it) {
        const int head = wave + 4 * it;
        const bool act = head < heads;   // wave-uniform
        f32x16 ctx;
#pragma unroll
        for (int i = 0; i < 16; ++i) ctx[i] = 0.0f;
        float zinv = 0.0f;
        if (act) {
            bf16x8 wk[CK], wv[CK], xa[CK];
            load_w(1, head, wk);
            // ---- pass 1
            float m = -__builtin_inff();
            for (int t0 = 0; t0 < n; t0 += 32) {
                load_x(t0, xa);
                f32x16 kk;
#pragma unroll
                for (int i = 0; i < 16; ++i) kk[i] = 0.0f;
#pragma unroll
                for (int s = 0; s < CK; ++s) kk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[s], wk[s], kk, 0, 0, 0);
                const bool full = t0 + 32 <= n;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (full || t0 + row_of(i) < n) m = fmaxf(m, kk[i]);
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            // ---- pass 2
            load_w(2, head, wv);
            float z = 0.0f;
            for (int t0 = 0; t0 < n; t0 += 32) {
                load_x(t0, xa);
                f32x16 kk, vv;
#pragma unroll
                for (int i = 0; i < 16; ++i) { kk[i] = 0.0f; vv[i] = 0.0f; }
#pragma unroll
                for (int s = 0; s < CK; ++s) {
                    kk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[s], wk[s], kk, 0, 0, 0);
                    vv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[s], wv[s], vv, 0, 0, 0);
                }
                const bool full = t0 + 32 <= n;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float e = __expf(kk[i] - m);
                    if (!full && t0 + row_of(i) >= n) e = 0.0f;   // (V of such a row is zero already: its x is)
                    z += e;
                    kk[i] = e;
                }
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag<0>(kk), acc_frag<0>(vv), ctx, 0, 0, 0);
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag<1>(kk), acc_frag<1>(vv), ctx, 0, 0, 0);
            }
            z += __shfl_xor(z, 32);
            zinv = 1.0f / (z * (float)n);
            if (h2 == 0) zs[wave][r] = zinv;
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < 16; ++i) ctx[i] *= zs[wave][row_of(i)];
            const bf16x8 c0 = acc_frag<0>(ctx), c1 = acc_frag<1>(ctx);
            bf16x8 wq[CK], xa[CK];
            load_w(0, head, wq);
            const float rs = 0.17677669529663687f;   // 1 / sqrt(32)
            // ---- pass 3
            for (int t0 = 0; t0 < n; t0 += 32) {
                load_x(t0, xa);
                f32x16 qt;
#pragma unroll
                for (int i = 0; i < 16; ++i) qt[i] = 0.0f;
#pragma unroll
                for (int s = 0; s < CK; ++s) qt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s], xa[s], qt, 0, 0, 0);
                float mx = qt[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = fmaxf(mx, qt[i]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    qt[i] = __expf(qt[i] - mx);
                    sum += qt[i];
                }
                sum += __shfl_xor(sum, 32);
                const float sc = rs / sum;
#pragma unroll
                for (int i = 0; i < 16; ++i) qt[i] *= sc;
                f32x16 o;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = 0.0f;
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, acc_frag<0>(qt), o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, acc_frag<1>(qt), o, 0, 0, 0);
                const int tok = t0 + r;
                if (tok < n) {
                    __hip_bfloat16* op = ob + (size_t)tok * HD + head * kHd + 4 * h2;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<uint2*>(op + 8 * c) =
                            make_uint2(pk_bf16(o[4 * c], o[4 * c + 1]), pk_bf16(o[4 * c + 2], o[4 * c + 3]));
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 3 x 3 convolution (stride 1, zero padding 1) on token-major bfloat16 activations as an implicit GEMM on the matrix
// cores, for the network's narrow layers (Cin = 64 or 128).  A workgroup keeps the weights of its 32 * NB output
// channels in LDS for its whole life -- [tap][k-step][lane half][channel] 16-byte chunks, so a wave's operand read is
// 32 consecutive chunks per half, conflict-free -- and walks tiles of 32 * NW output pixels (a wave = 32 pixels x NB
// accumulator tiles of 32 channels).  The tile's input is the flattened pixel range [p0 - W - 1, p0 + tile + W + 1):
// it is staged in LDS once (16-byte chunks, one pad chunk per pixel so that 32 consecutive pixels' chunks fall on
// distinct banks) and all nine taps read their fragments from there -- read straight from memory, nine times, the
// fragment loads ran at 6.5 TB/s of L2 traffic and were 2/3 of the kernel (tools/bench_conv.py, FBSMI_CONV_PROBE).  The
// next tile's chunks travel in registers while this tile is multiplied.  The product is formed transposed
// (out^T = W X^T: output channels in the registers, the pixel on the lane), so a lane stores 4 consecutive channels
// of its pixel at a time.  Zero padding: a tap whose neighbour is outside the image contributes a zero fragment (the
// flattened range holds some other pixel there).
// ------------------------------------------------------------------------------------------
#ifndef FBSMI_CONV_PROBE
#define FBSMI_CONV_PROBE 0   // diagnostic builds (tools/build_variants.sh): 1 no activation fragments, 2 no weight reads, 3 no input fetch
#endif
template <int CK, int NB, int NW, int MB>   // Cin = 16 * CK; 32 * NB output channels, 32 * MB pixels per wave, NW waves
__global__ void __launch_bounds__(64 * NW) k_conv3x3(const __hip_bfloat16* __restrict__ x, const __hip_bfloat16* __restrict__ w,
                                                     const float* __restrict__ bias, __hip_bfloat16* __restrict__ y, int H, int W,
                                                     int Cout, long long npix, long long ntiles, int xs, int ws, int ci0,
                                                     int accumulate) {
    extern __shared__ uint4 lds[];
    constexpr int Cin = 16 * CK, kThreads = 64 * NW, kTile = 32 * MB * NW, kNco = 32 * NB;
    constexpr int kPix = 2 * CK + 1;                 // 16-byte chunks per staged pixel (the last one is padding)
    uint4* wl = lds;                                 // [9][CK][2][kNco]
    uint4* patch = lds + 9 * CK * 2 * kNco;          // [kTile + 2 W + 2 (+ 1 zero pixel)][kPix]
    const int co0 = blockIdx.y * kNco;
    for (int i = threadIdx.x; i < 9 * CK * 2 * kNco; i += kThreads) {
        const int col = i % kNco, rest = i / kNco, hh = rest & 1, s = (rest >> 1) % CK, tap = (rest >> 1) / CK;
        wl[i] = *reinterpret_cast<const uint4*>(w + ((size_t)(co0 + col) * 9 + tap) * ws + ci0 + 16 * s + 8 * hh);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int npatch = kTile + 2 * W + 2, nchunk = npatch * 2 * CK;
    constexpr int kMaxPer = NW >= 6 ? 12 : 24;       // staged chunks per thread (host: nchunk <= kMaxPer * kThreads)
    // what does not change from tile to tile is worked out once: which chunk of the staged range a thread fetches (pixel
    // offset, element offset, LDS slot); pixel indices fit 32 bits (host check)
    int fdq[kMaxPer], foff[kMaxPer], fslot[kMaxPer];
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) {
        const int c = threadIdx.x + j * kThreads;
        const bool live = c < nchunk;
        fdq[j] = live ? c / (2 * CK) : -0x40000000;  // dead slots fail the range test below
        foff[j] = (c / (2 * CK)) * xs + 8 * (c % (2 * CK));
        fslot[j] = live ? (c / (2 * CK)) * kPix + c % (2 * CK) : -1;
    }
    const int np = (int)npix;
    uint4 stage[kMaxPer];
    auto fetch = [&](int t) {                        // tile t's input chunks -> registers (zeros outside the batch)
        const int q0 = t * kTile - W - 1;
        const __hip_bfloat16* xq = x + (long long)q0 * xs;   // wave-uniform base, 32-bit lane offsets
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j) {
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            if (FBSMI_CONV_PROBE != 3 && (unsigned)(q0 + fdq[j]) < (unsigned)np) u = *reinterpret_cast<const uint4*>(xq + foff[j]);
            stage[j] = u;
        }
    };
    // a tap whose neighbour is outside the image reads an all-zero pixel instead (one more slot behind the staged range)
    uint4* zero_px = patch + npatch * kPix;
    if (threadIdx.x < 2 * CK) zero_px[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    int t = blockIdx.x;
    const int nt = (int)ntiles, tstep = gridDim.x;
    if (t < nt) fetch(t);
    for (; t < nt; t += tstep) {
        __syncthreads();                             // the previous tile's reads are done (and the weights are in)
#pragma unroll
        for (int j = 0; j < kMaxPer; ++j)
            if (fslot[j] >= 0) patch[fslot[j]] = stage[j];
        __syncthreads();
        if (t + tstep < nt) fetch(t + tstep);
        const uint4* pcs[MB][9];                     // where this lane's fragments of tap (dy, dx) start: its neighbour, or zeros
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int p = t * kTile + 32 * (MB * wave + mb) + r;
            const bool pv = p < np;
            const unsigned pp = pv ? p : np - 1;
            const unsigned prow = pp / (unsigned)W;
            const int xw = (int)(pp - prow * W), yh = (int)(prow % (unsigned)H);
            const uint4* pc = patch + (32 * (MB * wave + mb) + r + W + 1) * kPix + h2;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool ok = pv && (unsigned)(yh + dy) < (unsigned)H && (unsigned)(xw + dx) < (unsigned)W;
                pcs[mb][tap] = ok ? pc + (dy * W + dx) * kPix : zero_px + h2;
            }
        }
        f32x16 acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;
        // k-steps (tap, s) in order; the fragments of step i + 2 are requested before the products of step i are issued
        // (scheduling barriers: left alone, the compiler reads each fragment right before its use and every pair of products
        // waits out an LDS round trip)
        constexpr int kSteps = 9 * CK, kAhead = 2;
        uint4 fa[kAhead + 1][MB], fw[kAhead + 1][NB];
        auto request = [&](int i, uint4 (&xa)[MB], uint4 (&wa)[NB]) {
            const int tap = i / CK, s = i % CK;
            const uint4* wp = wl + ((tap * CK + s) * 2 + h2) * kNco + r;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) xa[mb] = FBSMI_CONV_PROBE == 1 ? make_uint4(lane, tap, s, 0x3f803f80u) : pcs[mb][tap][2 * s];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wa[nb] = wp[32 * nb];
        };
#pragma unroll
        for (int i = 0; i < kAhead; ++i) request(i, fa[i], fw[i]);
#pragma unroll
        for (int i = 0; i < kSteps; ++i) {
            if (i + kAhead < kSteps) request(i + kAhead, fa[(i + kAhead) % (kAhead + 1)], fw[(i + kAhead) % (kAhead + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 xa = __builtin_bit_cast(bf16x8, fa[i % (kAhead + 1)][mb]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 wa = FBSMI_CONV_PROBE == 2 ? xa : __builtin_bit_cast(bf16x8, fw[i % (kAhead + 1)][nb]);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xa, acc[mb][nb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the tile's results leave as whole rows.  In the accumulators a lane holds 4-channel groups of ITS pixel, and
        // stored from there every lane of a store instruction writes 8 bytes of a different row (a write request each: that
        // was half of this kernel's time).  Instead: the lane halves swap the odd / even groups (v_permlane32_swap) so a
        // lane owns 16-byte chunks, the wave parks its 32 x (32 NB) tile in LDS -- in the staged input's memory, once every
        // wave is done reading it -- and reads it back with consecutive lanes along the rows.
        __syncthreads();
        constexpr int kRow = 2 * kNco + 16;                     // bytes of a parked row (one pad chunk)
        constexpr int kCpr = kNco / 8;                          // 16-byte chunks per row
        char* park = reinterpret_cast<char*>(patch) + (size_t)wave * MB * 32 * kRow;
        const float4* bp = reinterpret_cast<const float4*>(bias + co0 + 4 * h2);   // read only when bias != NULL
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            char* pk = park + (size_t)mb * 32 * kRow;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                    if (bias) { b0 = bp[8 * nb + 4 * j]; b1 = bp[8 * nb + 4 * j + 2]; }
                    const f32x16& A = acc[mb][nb];
                    const unsigned p0x = pk_bf16(A[8 * j] + b0.x, A[8 * j + 1] + b0.y), p0y = pk_bf16(A[8 * j + 2] + b0.z, A[8 * j + 3] + b0.w);
                    const unsigned p1x = pk_bf16(A[8 * j + 4] + b1.x, A[8 * j + 5] + b1.y), p1y = pk_bf16(A[8 * j + 6] + b1.z, A[8 * j + 7] + b1.w);
                    const auto rx = __builtin_amdgcn_permlane32_swap(p0x, p1x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(p0y, p1y, false, false);
                    *reinterpret_cast<uint4*>(pk + r * kRow + nb * 64 + (2 * j + h2) * 16) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                }
            // rows back out: lane l takes chunk l % kCpr of row l / kCpr (+ 64 / kCpr per round)
            const long long prow0 = (long long)t * kTile + 32 * (MB * wave + mb);
#pragma unroll
            for (int rd = 0; rd < kCpr / 2; ++rd) {
                const int row = lane / kCpr + rd * (64 / kCpr), ch = lane % kCpr;
                uint4 u = *reinterpret_cast<const uint4*>(pk + row * kRow + ch * 16);
                const long long pr = prow0 + row;
                if (pr < npix) {
                    uint4* dst = reinterpret_cast<uint4*>(y + pr * Cout + co0 + 8 * ch);
                    if (accumulate) {   // a later channel slice of the same convolution: add to what the earlier ones left
                        const uint4 o = *dst;
                        const unsigned ow[4] = {o.x, o.y, o.z, o.w}, uw[4] = {u.x, u.y, u.z, u.w};
                        unsigned rw[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            rw[k] = pk_bf16(__uint_as_float(ow[k] << 16) + __uint_as_float(uw[k] << 16),
                                            __uint_as_float(ow[k] & 0xffff0000u) + __uint_as_float(uw[k] & 0xffff0000u));
                        u = make_uint4(rw[0], rw[1], rw[2], rw[3]);
                    }
                    *dst = u;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 1x1 projection to 64 channels on the matrix cores, with what follows it in the network folded in:
//   y = [LN_c]( a Wa^T [+ b Wb^T] [+ bias] ) [* scale] [+ residual]
// a, b: token-major bfloat16 inputs of CKA * 16 / CKB * 16 channels whose concatenation the weight (64, Ka + Kb) multiplies
// (the up path's (h, skip) pairs: the concatenation is never formed); LN_c: the channel LayerNorm of LinearAttention's to_out
// (no bias, eps), `residual` the attention block's skip connection.  A wave takes 32 pixels: the product is formed
// transposed (channels in the registers, the pixel on the lane) from operand fragments read straight from memory (every
// activation element is read once: nothing to stage), the 64 channels of a pixel are then two accumulator tiles of ITS lane
// pair, so the LayerNorm statistics are in-lane sums plus one exchange; rows leave through an LDS park as in k_conv3x3.
// ------------------------------------------------------------------------------------------
template <int CKA, int CKB, bool LN>
__global__ void __launch_bounds__(256) k_proj64(const __hip_bfloat16* __restrict__ a, const __hip_bfloat16* __restrict__ b,
                                                const __hip_bfloat16* __restrict__ w, const float* __restrict__ bias,
                                                const float* __restrict__ scale, float eps,
                                                const __hip_bfloat16* __restrict__ residual, __hip_bfloat16* __restrict__ y,
                                                long long npix) {
    constexpr int CK = CKA + CKB, K = 16 * CK, kRow = 128 + 16;
    __shared__ __attribute__((aligned(16))) char park_all[4 * 32 * kRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    bf16x8 wf[2][CK];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < CK; ++s)
            wf[nb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(w + (size_t)(32 * nb + r) * K + 16 * s + 8 * h2));
    float bs[2][4][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) bs[nb][c][i] = bias ? bias[32 * nb + 8 * c + 4 * h2 + i] : 0.0f;
    char* park = park_all + wave * 32 * kRow;
    const long long ntile = (npix + 31) / 32;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < ntile; t += (long long)gridDim.x * 4) {
        const long long p = t * 32 + r;
        const bool pv = p < npix;
        const long long pp = pv ? p : npix - 1;
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;
        const uint4* pa = reinterpret_cast<const uint4*>(a + pp * (16 * CKA) + 8 * h2);
        uint4 xa[CK];
#pragma unroll
        for (int s = 0; s < CKA; ++s) xa[s] = pa[2 * s];
        if (CKB > 0) {
            const uint4* pb = reinterpret_cast<const uint4*>(b + pp * (16 * CKB) + 8 * h2);
#pragma unroll
            for (int s = 0; s < CKB; ++s) xa[CKA + s] = pb[2 * s];
        }
#pragma unroll
        for (int s = 0; s < CK; ++s) {
            const bf16x8 xf = __builtin_bit_cast(bf16x8, xa[s]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][s], xf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][s], xf, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] += bs[nb][i >> 2][i & 3];
        if (LN) {   // this pixel's 64 channels: 2 x 16 registers here, the other 32 on lane ^ 32
            float sm = 0.0f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sm += acc[nb][i];
            sm += __shfl_xor(sm, 32);
            const float mean = sm * (1.0f / 64.0f);
            float q = 0.0f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) q += (acc[nb][i] - mean) * (acc[nb][i] - mean);
            q += __shfl_xor(q, 32);
            const float rs = rsqrtf(q * (1.0f / 64.0f) + eps);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[nb][i] = (acc[nb][i] - mean) * rs * scale[32 * nb + 8 * (i >> 2) + 4 * h2 + (i & 3)];
        }
        // rows out (see k_conv3x3): 4-channel groups -> 16-byte chunks by swapping between the lane halves, parked, read back
        // along the rows; the residual joins there, with whole-row loads
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& A = acc[nb];
                const unsigned p0x = pk_bf16(A[8 * j], A[8 * j + 1]), p0y = pk_bf16(A[8 * j + 2], A[8 * j + 3]);
                const unsigned p1x = pk_bf16(A[8 * j + 4], A[8 * j + 5]), p1y = pk_bf16(A[8 * j + 6], A[8 * j + 7]);
                const auto rx = __builtin_amdgcn_permlane32_swap(p0x, p1x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(p0y, p1y, false, false);
                *reinterpret_cast<uint4*>(park + r * kRow + nb * 64 + (2 * j + h2) * 16) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int row = lane / 8 + rd * 8, ch = lane % 8;
            uint4 u = *reinterpret_cast<const uint4*>(park + row * kRow + ch * 16);
            const long long pr = t * 32 + row;
            if (pr < npix) {
                if (residual) {
                    const uint4 o = *reinterpret_cast<const uint4*>(residual + pr * 64 + 8 * ch);
                    const unsigned ow[4] = {o.x, o.y, o.z, o.w}, uw[4] = {u.x, u.y, u.z, u.w};
                    unsigned rw[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        rw[k] = pk_bf16(__uint_as_float(ow[k] << 16) + __uint_as_float(uw[k] << 16),
                                        __uint_as_float(ow[k] & 0xffff0000u) + __uint_as_float(uw[k] & 0xffff0000u));
                    u = make_uint4(rw[0], rw[1], rw[2], rw[3]);
                }
                *reinterpret_cast<uint4*>(y + pr * 64 + 8 * ch) = u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm + modulation + SiLU.  One workgroup per image; a thread owns one aligned 8-channel vector slot
// (always inside one group, since the group size is a multiple of 8) across a strided set of tokens:
// pass 1 Welford-merges its vectors into (count, mean, M2), the slots of a group are merged through LDS
// (Chan's formula); pass 2 re-reads (L2-hot), normalises, modulates, applies SiLU and stores.
// ------------------------------------------------------------------------------------------
struct Vec8 {
    float v[8];
};
__device__ __forceinline__ Vec8 ld8(const float* p) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    return Vec8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ Vec8 ld8(const __hip_bfloat16* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    Vec8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o.v[2 * i] = __uint_as_float(w[i] << 16);
        o.v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
    return o;
}
__device__ __forceinline__ void st8(float* p, const Vec8& x) {
    reinterpret_cast<float4*>(p)[0] = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ void st8(__hip_bfloat16* p, const Vec8& x) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __hip_bfloat16 lo = __float2bfloat16(x.v[2 * i]), hi = __float2bfloat16(x.v[2 * i + 1]);
        w[i] = (unsigned)(*reinterpret_cast<const unsigned short*>(&lo)) |
               ((unsigned)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T>
__global__ void __launch_bounds__(256) k_groupnorm_silu(const T* __restrict__ x, T* __restrict__ y, int n, int C, int groups,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ xbias,
                                                        const T* __restrict__ residual, const float* __restrict__ rbias) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int slots = C >> 3;                 // 8-channel vector slots per token (<= 256 / 8 ... C <= 2048)
    const int spg = slots / groups;           // slots per group
    const int lanes = 256 / slots;            // token lanes (slots divides 256: C in {64, 128, 256, 512, ...})
    const int slot = t % slots, tl = t / slots;
    const T* xb = x + (size_t)b * n * C + slot * 8;
    T* yb = y + (size_t)b * n * C + slot * 8;
    __shared__ float sN[256], sMean[256], sM2[256];
    __shared__ float gMean[32], gRstd[32];
    float xb8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xb8[i] = xbias ? xbias[slot * 8 + i] : 0.0f;
    // ---- pass 1
    float cnt = 0.0f, mean = 0.0f, M2 = 0.0f;
    if (tl < lanes) {
        auto merge = [&](Vec8 v) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += xb8[i];
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v.v[i];
            const float m8 = s * 0.125f;
            float q = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) q += (v.v[i] - m8) * (v.v[i] - m8);
            const float nn = cnt + 8.0f, dlt = m8 - mean;
            mean += dlt * (8.0f / nn);
            M2 += q + dlt * dlt * (cnt * 8.0f / nn);
            cnt = nn;
        };
        int tok = tl;
        for (; tok + 3 * lanes < n; tok += 4 * lanes) {   // four vectors in flight per thread (one workgroup streams an image)
            const Vec8 v0 = ld8(xb + (size_t)tok * C), v1 = ld8(xb + (size_t)(tok + lanes) * C),
                       v2 = ld8(xb + (size_t)(tok + 2 * lanes) * C), v3 = ld8(xb + (size_t)(tok + 3 * lanes) * C);
            merge(v0); merge(v1); merge(v2); merge(v3);
        }
        for (; tok < n; tok += lanes) merge(ld8(xb + (size_t)tok * C));
    }
    sN[t] = cnt;
    sMean[t] = mean;
    sM2[t] = M2;
    __syncthreads();
    if (t < groups) {
        float N = 0.0f, Mn = 0.0f, Mq = 0.0f;
        for (int l = 0; l < lanes; ++l)
            for (int sIdx = t * spg; sIdx < (t + 1) * spg; ++sIdx) {
                const int j = l * slots + sIdx;
                const float nb = sN[j];
                if (nb > 0.0f) {
                    const float nt = N + nb, dlt = sMean[j] - Mn;
                    Mn += dlt * (nb / nt);
                    Mq += sM2[j] + dlt * dlt * (N * nb / nt);
                    N = nt;
                }
            }
        gMean[t] = Mn;
        gRstd[t] = rsqrtf(Mq / N + eps);
    }
    __syncthreads();
    // ---- pass 2
    if (tl < lanes) {
        const int g = slot / spg;
        const float mu = gMean[g], rs = gRstd[g];
        float a[8], c[8];   // y = silu(x * a + c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = slot * 8 + i;
            const float ga = gamma[ch] * rs, be = beta[ch] - mu * ga;
            const float sc = scale ? 1.0f + scale[(size_t)b * C + ch] : 1.0f, sh = shift ? shift[(size_t)b * C + ch] : 0.0f;
            a[i] = ga * sc;
            c[i] = fmaf(xb8[i], a[i], be * sc + sh);   // (x + xbias) * a + c0
        }
        const T* rb = residual ? residual + (size_t)b * n * C + slot * 8 : nullptr;
        float rb8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rb8[i] = (rb && rbias) ? rbias[slot * 8 + i] : 0.0f;
        auto apply = [&](int tok, Vec8 v, const Vec8& r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float z = fmaf(v.v[i], a[i], c[i]);
                v.v[i] = z / (1.0f + expf(-z));
            }
            if (rb)
#pragma unroll
                for (int i = 0; i < 8; ++i) v.v[i] += r.v[i] + rb8[i];
            st8(yb + (size_t)tok * C, v);
        };
        int tok = tl;
        for (; tok + 3 * lanes < n; tok += 4 * lanes) {
            Vec8 v[4], r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = ld8(xb + (size_t)(tok + j * lanes) * C);
                if (rb) r[j] = ld8(rb + (size_t)(tok + j * lanes) * C);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) apply(tok + j * lanes, v[j], r[j]);
        }
        for (; tok < n; tok += lanes) {
            Vec8 r;
            if (rb) r = ld8(rb + (size_t)tok * C);
            apply(tok, ld8(xb + (size_t)tok * C), r);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Channel LayerNorm: a row of C channels is C / 8 adjacent lanes (one 16-byte vector each), reductions by
// xor-shuffles inside that lane group; one read, one write.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_channel_layernorm(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int C,
                                                           const float* __restrict__ scale, float eps,
                                                           const T* __restrict__ residual, const float* __restrict__ xbias) {
    const int L = C >> 3;                                   // lanes per row (power of two, <= 64)
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L;
    const int slot = threadIdx.x % L;
    const bool live = row < rows;
    Vec8 v;
    if (live) {
        v = ld8(x + row * C + slot * 8);
        if (xbias)
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += xbias[slot * 8 + i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v.v[i] = 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v.v[i];
    for (int m = 1; m < L; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (v.v[i] - mean) * (v.v[i] - mean);
    for (int m = 1; m < L; m <<= 1) q += __shfl_xor(q, m);
    const float rs = rsqrtf(q / (float)C + eps);
    if (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v.v[i] = (v.v[i] - mean) * rs * scale[slot * 8 + i];
        if (residual) {
            const Vec8 r = ld8(residual + row * C + slot * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) v.v[i] += r.v[i];
        }
        st8(y + row * C + slot * 8, v);
    }
}

// ------------------------------------------------------------------------------------------
// y[r][c] += bias[c] in place on (rows, C) token-major activations, 8 channels (16 bytes of bfloat16) per thread: what
// torch's convolution does with a broadcasting elementwise kernel at a fraction of the memory rate.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_bias_add(T* __restrict__ y, const float* __restrict__ bias, int64_t vecs, int C8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= vecs) return;
    const int slot = (int)(i % C8);
    Vec8 v = ld8(y + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] += bias[slot * 8 + k];
    st8(y + i * 8, v);
}

// einops 'b h w (h2 w2 c) -> b (h h2) (w w2) c' (fbs/nn/utils.py:53-57) on token-major activations, with the bias of the
// convolution that produced x added on the way: x (B, H, W, s*s*c) -> y (B, s*H, s*W, c); a thread moves 8 channels.
template <typename T>
__global__ void __launch_bounds__(256) k_pixel_shuffle(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ bias,
                                                       int64_t vecs, int H, int W, int c8, int s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // output vector index: (((b, oy), ox), cv)
    if (i >= vecs) return;
    const int cv = (int)(i % c8);
    int64_t r = i / c8;
    const int ox = (int)(r % (W * s));
    r /= (W * s);
    const int oy = (int)(r % (H * s));
    const int64_t b = r / (H * s);
    const int h2 = oy % s, w2 = ox % s;
    const int cin = ((h2 * s + w2) * c8 + cv) * 8;
    Vec8 v = ld8(x + (((b * H + oy / s) * W + ox / s) * (int64_t)(s * s * c8)) * 8 + cin);
    if (bias)
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] += bias[cin + k];
    st8(y + i * 8, v);
}

}  // namespace fbsmi

using namespace fbsmi;

extern "C" int fbsmi_nn_linear_attention(const void* qkv, void* out, int dtype, int64_t B, int32_t n, int32_t heads,
                                         int32_t dim_head, void* stream) {
    if (!qkv || !out || B < 0 || n < 1 || heads < 1 || (dtype != 0 && dtype != 1))
        return fail(FBSMI_ERR_ARG, "nn_linear_attention: bad arguments");
    if (dim_head != kHd) return fail(FBSMI_ERR_UNSUPPORTED, "nn_linear_attention: dim_head must be 32");
    if (B == 0) return FBSMI_OK;
    if (B > 65535) return fail(FBSMI_ERR_UNSUPPORTED, "nn_linear_attention: more than 65535 images per call");
    const dim3 grid(heads, (unsigned)B);
    if (dtype == 0)
        k_linear_attention<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)qkv, (float*)out, n, heads);
    else
        k_linear_attention<__hip_bfloat16><<<grid, 256, 0, (hipStream_t)stream>>>((const __hip_bfloat16*)qkv,
                                                                                 (__hip_bfloat16*)out, n, heads);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_groupnorm_silu(const void* x, void* y, int dtype, int64_t B, int32_t n, int32_t C, int32_t groups,
                                       const float* gamma, const float* beta, float eps, const float* scale,
                                       const float* shift, const float* xbias, const void* residual, const float* rbias,
                                       void* stream) {
    if (!x || !y || !gamma || !beta || B < 0 || n < 1 || C < 8 || groups < 1 || groups > 32 || (dtype != 0 && dtype != 1) ||
        (scale == nullptr) != (shift == nullptr))
        return fail(FBSMI_ERR_ARG, "nn_groupnorm_silu: bad arguments");
    const int slots = C / 8;
    if (C % (8 * groups) != 0 || slots > 256 || 256 % slots != 0)
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_groupnorm_silu: C must be a multiple of 8 * groups and C / 8 must divide 256");
    if (B == 0) return FBSMI_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        k_groupnorm_silu<float><<<(unsigned)B, 256, 0, st>>>((const float*)x, (float*)y, n, C, groups, gamma, beta, eps,
                                                            scale, shift, xbias, (const float*)residual, rbias);
    else
        k_groupnorm_silu<__hip_bfloat16><<<(unsigned)B, 256, 0, st>>>((const __hip_bfloat16*)x, (__hip_bfloat16*)y, n, C,
                                                                     groups, gamma, beta, eps, scale, shift, xbias,
                                                                     (const __hip_bfloat16*)residual, rbias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_channel_layernorm(const void* x, void* y, int dtype, int64_t rows, int32_t C, const float* scale,
                                          float eps, const void* residual, const float* xbias, void* stream) {
    if (!x || !y || !scale || rows < 0 || C < 8 || (dtype != 0 && dtype != 1))
        return fail(FBSMI_ERR_ARG, "nn_channel_layernorm: bad arguments");
    const int L = C / 8;
    if (C % 8 != 0 || L > 64 || (L & (L - 1)) != 0)
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_channel_layernorm: C / 8 must be a power of two <= 64");
    if (rows == 0) return FBSMI_OK;
    const int64_t per_block = 256 / L;
    const int64_t blocks = (rows + per_block - 1) / per_block;
    if (blocks > 0x7fffffff) return fail(FBSMI_ERR_UNSUPPORTED, "nn_channel_layernorm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        k_channel_layernorm<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)x, (float*)y, rows, C, scale, eps,
                                                                     (const float*)residual, xbias);
    else
        k_channel_layernorm<__hip_bfloat16><<<(unsigned)blocks, 256, 0, st>>>((const __hip_bfloat16*)x, (__hip_bfloat16*)y,
                                                                             rows, C, scale, eps, (const __hip_bfloat16*)residual, xbias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_bias_add(void* y, int dtype, int64_t rows, int32_t C, const float* bias, void* stream) {
    if (!y || !bias || rows < 0 || C < 8 || C % 8 != 0 || (dtype != 0 && dtype != 1))
        return fail(FBSMI_ERR_ARG, "nn_bias_add: bad arguments (C must be a multiple of 8)");
    const int64_t vecs = rows * (C / 8), blocks = (vecs + 255) / 256;
    if (vecs == 0) return FBSMI_OK;
    if (blocks > 0x7fffffff) return fail(FBSMI_ERR_UNSUPPORTED, "nn_bias_add: too many elements");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) k_bias_add<float><<<(unsigned)blocks, 256, 0, st>>>((float*)y, bias, vecs, C / 8);
    else k_bias_add<__hip_bfloat16><<<(unsigned)blocks, 256, 0, st>>>((__hip_bfloat16*)y, bias, vecs, C / 8);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_pixel_shuffle(const void* x, void* y, int dtype, int64_t B, int32_t H, int32_t W, int32_t c, int32_t s,
                                      const float* bias, void* stream) {
    if (!x || !y || B < 0 || H < 1 || W < 1 || c < 8 || c % 8 != 0 || s < 1 || (dtype != 0 && dtype != 1))
        return fail(FBSMI_ERR_ARG, "nn_pixel_shuffle: bad arguments (c must be a multiple of 8)");
    const int64_t vecs = B * H * s * W * s * (c / 8), blocks = (vecs + 255) / 256;
    if (vecs == 0) return FBSMI_OK;
    if (blocks > 0x7fffffff) return fail(FBSMI_ERR_UNSUPPORTED, "nn_pixel_shuffle: too many elements");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) k_pixel_shuffle<float><<<(unsigned)blocks, 256, 0, st>>>((const float*)x, (float*)y, bias, vecs, H, W, c / 8, s);
    else
        k_pixel_shuffle<__hip_bfloat16><<<(unsigned)blocks, 256, 0, st>>>((const __hip_bfloat16*)x, (__hip_bfloat16*)y, bias, vecs,
                                                                         H, W, c / 8, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_qkv_linear_attention(const void* xn, const void* w, void* out, int64_t B, int32_t n, int32_t C,
                                             int32_t heads, int32_t dim_head, void* stream) {
    if (!xn || !w || !out || B < 0 || n < 1 || heads < 1) return fail(FBSMI_ERR_ARG, "nn_qkv_linear_attention: bad arguments");
    if (dim_head != kHd) return fail(FBSMI_ERR_UNSUPPORTED, "nn_qkv_linear_attention: dim_head must be 32");
    if (C != 16 && C != 32 && C != 64 && C != 128)
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_qkv_linear_attention: C must be 16, 32, 64 or 128");
    if (B == 0) return FBSMI_OK;
    if (B > 0x7fffffff) return fail(FBSMI_ERR_UNSUPPORTED, "nn_qkv_linear_attention: too many images");
    hipStream_t st = (hipStream_t)stream;
    const __hip_bfloat16 *x_ = (const __hip_bfloat16*)xn, *w_ = (const __hip_bfloat16*)w;
    __hip_bfloat16* o_ = (__hip_bfloat16*)out;
    switch (C) {
        case 16: k_qkv_linear_attention<1><<<(unsigned)B, 256, 0, st>>>(x_, w_, o_, n, heads); break;
        case 32: k_qkv_linear_attention<2><<<(unsigned)B, 256, 0, st>>>(x_, w_, o_, n, heads); break;
        case 64: k_qkv_linear_attention<4><<<(unsigned)B, 256, 0, st>>>(x_, w_, o_, n, heads); break;
        default: k_qkv_linear_attention<8><<<(unsigned)B, 256, 0, st>>>(x_, w_, o_, n, heads); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

namespace {
// Tile shape of k_conv3x3 for rows of W pixels and slices of Cin channels: 8 waves per workgroup when the staged range fits
// beside the weights (72 KB) in 160 KB of LDS, else 6, else 4; a wave multiplies 32 pixels (or 64: two accumulator rows
// sharing every weight fragment).  FBSMI_CONV_CFG=<waves><pixel blocks> overrides (diagnostic).  false: rows too wide.
struct ConvShape { int nw, mb; size_t lds; };
bool conv3x3_shape(int W, int Cin, ConvShape& out) {
    const int ck = Cin / 16;
    const int nco = ck == 4 ? 64 : 32;               // output channels per workgroup: 72 KB of weights either way
    auto lds_of = [&](int nw, int mb) { return (size_t)16 * (9 * ck * 2 * nco + (size_t)(32 * mb * nw + 2 * W + 3) * (2 * ck + 1)); };
    auto fits = [&](int nw, int mb) {
        return lds_of(nw, mb) <= 160 * 1024 &&
               (long long)(32 * mb * nw + 2 * W + 2) * 2 * ck <= (nw >= 6 ? 12ll : 24ll) * 64 * nw;
    };
    int nw = 8, mb = 1;      // measured (tools/bench_conv.py): 8 waves x 32 pixels beats 8 x 64 and both 4-wave shapes
    if (!fits(nw, mb)) nw = 6;   // (128-channel slices on 32-pixel rows miss the 8-wave tile by 1.3 KB of LDS)
    if (!fits(nw, mb)) nw = 4;
    if (const char* cfg = getenv("FBSMI_CONV_CFG")) { nw = cfg[0] - '0'; mb = cfg[1] - '0'; }
    if ((nw != 4 && nw != 6 && nw != 8) || (mb != 1 && mb != 2) || (nw == 6 && mb != 1) || !fits(nw, mb)) return false;
    out = ConvShape{nw, mb, lds_of(nw, mb)};
    return true;
}
}  // namespace

extern "C" int fbsmi_nn_conv3x3_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (H < 1 || W < 1 || (Cin != 64 && Cin != 128) || Cout < 64 || Cout % 64 != 0) return 0;
    ConvShape sh;
    return conv3x3_shape(W, Cin, sh) ? 1 : 0;
}

extern "C" int fbsmi_nn_conv3x3(const void* x, int32_t xstride, const void* w, int32_t wstride, int32_t ci_off, const float* bias,
                                void* y, int accumulate, int64_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream) {
    if (!x || !w || !y || B < 0 || H < 1 || W < 1 || xstride < Cin || xstride % 8 != 0 || ci_off < 0 || ci_off % 8 != 0 ||
        wstride < ci_off + Cin || wstride % 8 != 0)
        return fail(FBSMI_ERR_ARG, "nn_conv3x3: bad arguments");
    if ((Cin != 64 && Cin != 128) || Cout < 64 || Cout % 64 != 0)
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_conv3x3: Cin must be 64 or 128 and Cout a multiple of 64");
    if (B == 0) return FBSMI_OK;
    const long long npix = (long long)B * H * W;
    if (npix > 0x3fffffff) return fail(FBSMI_ERR_UNSUPPORTED, "nn_conv3x3: more than 2^30 pixels per call");
    const int ck = Cin / 16;
    const int nco = ck == 4 ? 64 : 32;
    ConvShape sh;
    if (!conv3x3_shape(W, Cin, sh))
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_conv3x3: image rows too wide for the staged range (ask fbsmi_nn_conv3x3_supported first)");
    const int nw = sh.nw, mb = sh.mb;
    const size_t lds = sh.lds;
    const int tile = 32 * mb * nw;
    const long long ntiles = (npix + tile - 1) / tile;
    const long long cap = 256 * (long long)((160 * 1024) / lds);   // workgroups resident at once
    const dim3 grid((unsigned)(ntiles < cap ? ntiles : cap), (unsigned)(Cout / nco));
    hipStream_t st = (hipStream_t)stream;
    const __hip_bfloat16 *x_ = (const __hip_bfloat16*)x, *w_ = (const __hip_bfloat16*)w;
    __hip_bfloat16* y_ = (__hip_bfloat16*)y;
    hipError_t e = hipSuccess;
#define FBSMI_CONV_LAUNCH(CK_, NB_, NW_, MB_)                                                                              \
    {                                                                                                                      \
        e = hipFuncSetAttribute((const void*)k_conv3x3<CK_, NB_, NW_, MB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) k_conv3x3<CK_, NB_, NW_, MB_><<<grid, 64 * NW_, lds, st>>>(x_, w_, bias, y_, H, W, Cout, npix, ntiles, xstride, wstride, ci_off, accumulate); \
    }
    if (ck == 4) {
        if (nw == 8 && mb == 2) FBSMI_CONV_LAUNCH(4, 2, 8, 2)
        else if (nw == 8) FBSMI_CONV_LAUNCH(4, 2, 8, 1)
        else if (nw == 6) FBSMI_CONV_LAUNCH(4, 2, 6, 1)
        else if (mb == 2) FBSMI_CONV_LAUNCH(4, 2, 4, 2)
        else FBSMI_CONV_LAUNCH(4, 2, 4, 1)
    } else {
        if (nw == 8 && mb == 2) FBSMI_CONV_LAUNCH(8, 1, 8, 2)
        else if (nw == 8) FBSMI_CONV_LAUNCH(8, 1, 8, 1)
        else if (nw == 6) FBSMI_CONV_LAUNCH(8, 1, 6, 1)
        else if (mb == 2) FBSMI_CONV_LAUNCH(8, 1, 4, 2)
        else FBSMI_CONV_LAUNCH(8, 1, 4, 1)
    }
#undef FBSMI_CONV_LAUNCH
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}

extern "C" int fbsmi_nn_proj64(const void* a, int32_t Ca, const void* b, int32_t Cb, const void* w, const float* bias,
                               const float* ln_scale, float eps, const void* residual, void* y, int64_t npix, void* stream) {
    if (!a || !w || !y || npix < 0 || (Cb > 0 && !b)) return fail(FBSMI_ERR_ARG, "nn_proj64: bad arguments");
    if (!((Ca == 64 || Ca == 128) && (Cb == 0 || Cb == 64)) || (Ca == 128 && Cb != 0))
        return fail(FBSMI_ERR_UNSUPPORTED, "nn_proj64: inputs of (64), (128) or (64, 64) channels");
    if (npix == 0) return FBSMI_OK;
    const long long tiles = (npix + 31) / 32, wgs = (tiles + 3) / 4;
    const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);
    hipStream_t st = (hipStream_t)stream;
    const __hip_bfloat16 *a_ = (const __hip_bfloat16*)a, *b_ = (const __hip_bfloat16*)b, *w_ = (const __hip_bfloat16*)w,
                         *r_ = (const __hip_bfloat16*)residual;
    __hip_bfloat16* y_ = (__hip_bfloat16*)y;
#define FBSMI_PROJ(CKA_, CKB_)                                                                                   \
    {                                                                                                            \
        if (ln_scale) k_proj64<CKA_, CKB_, true><<<grid, 256, 0, st>>>(a_, b_, w_, bias, ln_scale, eps, r_, y_, npix);  \
        else k_proj64<CKA_, CKB_, false><<<grid, 256, 0, st>>>(a_, b_, w_, bias, ln_scale, eps, r_, y_, npix);          \
    }
    if (Ca == 128) FBSMI_PROJ(8, 0)
    else if (Cb == 64) FBSMI_PROJ(4, 4)
    else FBSMI_PROJ(4, 0)
#undef FBSMI_PROJ
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e));
    return FBSMI_OK;
}
