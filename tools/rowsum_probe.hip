// How fast can one workgroup per row stream (parts of) a (2048, 12288) float matrix from cold HBM?  Variants of the
// log-density role's access pattern.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/rowsum_probe tools/rowsum_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float wsum(float v) {
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// WHAT 0: image order, every element, float4, all loads of a thread issued before the adds (12 per thread)
// WHAT 1: through the offset table (dv elements), wave w takes segments w, w+4, ...; loop not unrolled
// WHAT 2: as 1 but unrolled by 3 (three table loads, then three data loads)
// WHAT 3: as 1, plus the two L2-resident vectors (target, base)
// WHAT 4: image order but only the observed 75 % (skip by role < 0 test on first element; wide loads)
template <int WHAT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k(const float* __restrict__ net, const int32_t* __restrict__ v_off, const int32_t* __restrict__ role,
                                           const float* __restrict__ v, const float* __restrict__ vp, int D, int dv, float* out) {
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BLOCK / 64;
    const float* __restrict__ row = net + (int64_t)r * D;
    float acc = 0.f;
    if (WHAT == 0) {
        float4 x[12];
        const int per = D / 4 / BLOCK;  // 12 for 256 threads
#pragma unroll
        for (int i = 0; i < 12; ++i) if (i < per) x[i] = *(const float4*)(row + (i * BLOCK + threadIdx.x) * 4);
#pragma unroll
        for (int i = 0; i < 12; ++i) if (i < per) acc += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    } else if (WHAT == 4) {
        const int per = D / 4 / BLOCK;
        for (int i = 0; i < per; ++i) {
            const int e = (i * BLOCK + threadIdx.x) * 4;
            if (role[e] < 0) { const float4 x = *(const float4*)(row + e); acc += (x.x + x.y) + (x.z + x.w); }
        }
    } else {
        const int nseg = dv / 256;
        if (WHAT == 2) {
            for (int sg = wave; sg < nseg; sg += NW * 3) {
                int4 o[3]; f4u x[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) { const int s2 = sg + b * NW; o[b] = *(const int4*)(v_off + (s2 < nseg ? s2 : sg) * 256 + lane * 4); }
#pragma unroll
                for (int b = 0; b < 3; ++b) x[b] = *(const f4u*)(row + o[b].x);
#pragma unroll
                for (int b = 0; b < 3; ++b) if (sg + b * NW < nseg) acc += (x[b].x + x[b].y) + (x[b].z + x[b].w);
            }
        } else {
            for (int sg = wave; sg < nseg; sg += NW) {
                const int j0 = sg * 256 + lane * 4;
                const int4 o = *(const int4*)(v_off + j0);
                const f4u x = *(const f4u*)(row + o.x);
                acc += (x.x + x.y) + (x.z + x.w);
                if (WHAT == 3) {
                    const f4u a = *(const f4u*)(v + j0), b = *(const f4u*)(vp + j0);
                    acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
                }
            }
        }
    }
    acc = wsum(acc);
    __shared__ float sh[16];
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < NW; ++i) t += sh[i]; out[r] = t; }
}

int main() {
    const int n = 2048, W = 64, C = 3, S = 32, shift = 7;
    const int D = W * W * C, du = S * S * C, dv = D - du, NS = 6;
    std::vector<int32_t> role(D), voff(dv);
    { int p = 0, q = 0; std::vector<int> isu(W * W, 0);
      for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) isu[(shift + i) * W + shift + j] = 1;
      for (int px = 0; px < W * W; ++px) for (int ch = 0; ch < C; ++ch) { if (isu[px]) role[px * C + ch] = p++; else { voff[q] = px * C + ch; role[px * C + ch] = ~(q++); } } }
    int32_t *d_role, *d_voff; hipMalloc(&d_role, D * 4); hipMemcpy(d_role, role.data(), D * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_voff, dv * 4); hipMemcpy(d_voff, voff.data(), dv * 4, hipMemcpyHostToDevice);
    float *net[NS], *out, *v, *vp;
    hipMalloc(&out, n * 4); hipMalloc(&v, dv * 4); hipMalloc(&vp, dv * 4); hipMemset(v, 0, dv * 4); hipMemset(vp, 0, dv * 4);
    for (int s = 0; s < NS; ++s) { hipMalloc(&net[s], (size_t)n * D * 4); hipMemset(net[s], 0, (size_t)n * D * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double mb, auto launch) {
        for (int s = 0; s < NS; ++s) launch(s);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) for (int s = 0; s < NS; ++s) launch(s);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (10 * NS);
        printf("%-64s %.1f us  %.2f TB/s\n", name, us, mb / us * 1e-6);
    };
    const double all = (double)n * D * 4, obs = (double)n * dv * 4;
#define L(WHAT, BLOCK) [&](int s) { k<WHAT, BLOCK><<<n, BLOCK>>>(net[s], d_voff, d_role, v, vp, D, dv, out); }
    run("image order, all elements, 256 threads, 12 float4 in flight", all, L(0, 256));
    run("image order, all elements, 512 threads, 6 float4 in flight", all, L(0, 512));
    run("image order, all elements, 1024 threads, 3 float4 in flight", all, L(0, 1024));
    run("offset table, observed 75 %, 256 threads, 1 segment at a time", obs, L(1, 256));
    run("offset table, observed 75 %, 256 threads, 3 segments at a time", obs, L(2, 256));
    run("offset table, observed 75 %, 512 threads, 1 segment at a time", obs, L(1, 512));
    run("offset table, observed 75 %, 1024 threads, 1 segment at a time", obs, L(1, 1024));
    run("offset table + target + base, 256 threads", obs, L(3, 256));
    run("image order, observed 75 % by role test, 256 threads", obs, L(4, 256));
    return 0;
}
