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
