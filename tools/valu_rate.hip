// Vector-instruction issue rate on gfx950: how many cycles a SIMD needs per wave64 v_fma_f32 and per
// v_pk_fma_f32 at 1, 2, 4 and 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[8];
    f2 y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));  // keeps the SLP vectoriser from packing it
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += PK ? y[i].x + y[i].y : x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 1 wave per SIMD of a CU
            const int blocks = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) k<1><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f); else k<0><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 32 * wps;   // wave-instructions issued on one SIMD
            printf("%s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4,
                   (double)blocks * 256 * iters * 32 * (pk ? 4 : 2) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
