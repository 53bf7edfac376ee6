// Where does k_em_concat's time go?  Variants of the same kernel with parts removed, timed on cold buffers.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/concat_probe tools/concat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// WHAT: 0 full, 1 no gathers (store the role word), 2 gather only from rows (v_prev elements -> 0), 3 gather only from v_prev,
// 4 one dword store per lane instead of float4, 5 full but row-contiguous reads (role ignored: x = row[e % du])
template <int WHAT, int G>
__global__ void __launch_bounds__(256) k(const float* __restrict__ us, const int32_t* __restrict__ A, const float* __restrict__ v_prev,
                                         const int32_t* __restrict__ role, int32_t du, int32_t D, int32_t chunks, float* img) {
    const int32_t r = blockIdx.x / chunks, c = blockIdx.x - r * chunks;
    const int base = c * (1024 * G) + threadIdx.x * 4;
    const float* __restrict__ row = us + (int64_t)(A ? A[r] : r) * du;
    int4 ro[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (base + g * 1024 < D) ro[g] = *(const int4*)(role + base + g * 1024);
    float x[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (base + g * 1024 < D) {
            const int o[4] = {ro[g].x, ro[g].y, ro[g].z, ro[g].w};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                if (WHAT == 1) x[g][k2] = __int_as_float(o[k2]);
                else if (WHAT == 2) x[g][k2] = o[k2] >= 0 ? row[o[k2]] : 0.0f;
                else if (WHAT == 3) x[g][k2] = o[k2] >= 0 ? 1.0f : v_prev[~o[k2]];
                else if (WHAT == 5) x[g][k2] = row[(base + g * 1024 + k2) % du];
                else { const float* src = o[k2] >= 0 ? row + o[k2] : v_prev + ~o[k2]; x[g][k2] = *src; }
            }
        }
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (base + g * 1024 < D) {
            const int64_t at = (int64_t)r * D + base + g * 1024;
            *(float4*)(img + at) = make_float4(x[g][0], x[g][1], x[g][2], x[g][3]);
        }
}

// lane-contiguous dword accesses: element e = base + k * 256 + tid
template <int K>
__global__ void __launch_bounds__(256) kd(const float* __restrict__ us, const int32_t* __restrict__ A, const float* __restrict__ v_prev,
                                          const int32_t* __restrict__ role, int32_t du, int32_t D, int32_t chunks, float* img) {
    const int32_t r = blockIdx.x / chunks, c = blockIdx.x - r * chunks;
    const int base = c * (256 * K) + threadIdx.x;
    const float* __restrict__ row = us + (int64_t)(A ? A[r] : r) * du;
    int ro[K];
#pragma unroll
    for (int g = 0; g < K; ++g) if (base + g * 256 < D) ro[g] = role[base + g * 256];
    float x[K];
#pragma unroll
    for (int g = 0; g < K; ++g) if (base + g * 256 < D) { const float* src = ro[g] >= 0 ? row + ro[g] : v_prev + ~ro[g]; x[g] = *src; }
#pragma unroll
    for (int g = 0; g < K; ++g) if (base + g * 256 < D) img[(int64_t)r * D + base + g * 256] = x[g];
}

int main() {
    const int n = 2048, W = 64, C = 3, S = 32, shift = 7;
    const int D = W * W * C, du = S * S * C, dv = D - du, NS = 6;
    std::vector<int32_t> role(D);
    { int p = 0, q = 0;
      std::vector<int> isu(W * W, 0);
      for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) isu[(shift + i) * W + shift + j] = 1;
      // unobserved in (i, j) product order == ascending pixel order for a rectangle; observed ascending
      for (int px = 0; px < W * W; ++px) for (int ch = 0; ch < C; ++ch) role[px * C + ch] = isu[px] ? (p++) : ~(q++);
    }
    int32_t* d_role; hipMalloc(&d_role, D * 4); hipMemcpy(d_role, role.data(), D * 4, hipMemcpyHostToDevice);
    float *us[NS], *img[NS], *vp; int32_t* A[NS];
    hipMalloc(&vp, dv * 4); hipMemset(vp, 0, dv * 4);
    std::vector<int32_t> hA(n);
    for (int s = 0; s < NS; ++s) {
        hipMalloc(&us[s], (size_t)n * du * 4); hipMemset(us[s], 0, (size_t)n * du * 4);
        hipMalloc(&img[s], (size_t)n * D * 4);
        hipMalloc(&A[s], n * 4);
        for (int i = 0; i < n; ++i) hA[i] = (i * 7919 + s * 13) % n;
        hipMemcpy(A[s], hA.data(), n * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int s = 0; s < NS; ++s) launch(s);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) for (int s = 0; s < NS; ++s) launch(s);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.1f us\n", name, ms * 1e3 / (10 * NS));
    };
#define L(WHAT, G) [&](int s) { const int chunks = (D + 1024 * G - 1) / (1024 * G); k<WHAT, G><<<n * chunks, 256>>>(us[s], A[s], vp, d_role, du, D, chunks, img[s]); }
    run("full, 4 groups/thread", L(0, 4));
    run("full, 1 group/thread", L(0, 1));
    run("full, 2 groups/thread", L(0, 2));
    run("no gathers (store role words), 4 groups", L(1, 4));
    run("row gathers only, 4 groups", L(2, 4));
    run("v_prev gathers only, 4 groups", L(3, 4));
    run("row reads contiguous (no role), 4 groups", L(5, 4));
#define LD(K) [&](int s) { const int chunks = (D + 256 * K - 1) / (256 * K); kd<K><<<n * chunks, 256>>>(us[s], A[s], vp, d_role, du, D, chunks, img[s]); }
    run("dword lane-contiguous, 4 per thread", LD(4));
    run("dword lane-contiguous, 8 per thread", LD(8));
    run("dword lane-contiguous, 16 per thread", LD(16));
    run("hipMemsetAsync 100 MB", [&](int s) { hipMemsetAsync(img[s], 0, (size_t)n * D * 4, 0); });
    return 0;
}
