// Issue cost of the integer instructions Threefry-2x32 is made of, and of whole Threefry calls, on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o tools/int_rate tools/int_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../fbs_amd/csrc/fbsmi_device.h"

// OP: 0 v_add_u32, 1 v_alignbit_b32, 2 v_xor_b32, 3 v_xad_u32, 4 v_fma_f32, 5 v_mul_f32, 6 v_add_f32, 7 v_cndmask (via select), 8 v_rcp_f32
template <int OP>
__global__ void __launch_bounds__(256) k_op(uint32_t* out, int iters, uint32_t a) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x[i]));
                if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 3) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 11) asm volatile("v_lshrrev_b32 %0, 9, %0" : "+v"(x[i]));
                if (OP == 12) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCALL>
__global__ void __launch_bounds__(256) k_tf(uint32_t* out, int iters, uint32_t k0, uint32_t k1) {
    uint32_t c[NCALL], acc = 0;
    for (int i = 0; i < NCALL; ++i) c[i] = (blockIdx.x * 256 + threadIdx.x) * NCALL + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCALL; ++i) {
            uint32_t o0, o1;
            fbsmi::threefry2x32(k0, k1, c[i], c[i] + 12345u + it, o0, o1);
            acc ^= o0 + o1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
static double run(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t));
    const int iters = 20000, blocks = 256 * 8;
    const char* names[] = {"v_add_u32", "v_alignbit_b32", "v_xor_b32", "v_xad_u32", "v_fma_f32", "v_mul_f32", "v_add_f32", "", "v_rcp_f32",
                           "v_add3_u32", "v_lshl_add_u32", "v_lshrrev_b32", "v_or_b32"};
#define RUN_OP(OP) { double ms = run([&] { k_op<OP><<<blocks, 256>>>(out, iters, 3u); }); \
    printf("%-16s 8 waves/SIMD: %.2f cycles@2.4GHz per wave-instruction per SIMD\n", names[OP], ms * 1e6 / ((double)iters * 32 * 8) * 2.4); }
    RUN_OP(0) RUN_OP(1) RUN_OP(2) RUN_OP(3) RUN_OP(4) RUN_OP(5) RUN_OP(6) RUN_OP(8) RUN_OP(9) RUN_OP(10) RUN_OP(11) RUN_OP(12)
    const int it2 = 2000;
    { double ms = run([&] { k_tf<1><<<blocks, 256>>>(out, it2, 1u, 2u); });
      printf("threefry2x32 x1 per thread: %.1f cycles@2.4GHz per call per SIMD (8 waves/SIMD)\n", ms * 1e6 / ((double)it2 * 1 * 8) * 2.4); }
    { double ms = run([&] { k_tf<4><<<blocks, 256>>>(out, it2, 1u, 2u); });
      printf("threefry2x32 x4 per thread: %.1f cycles@2.4GHz per call per SIMD (8 waves/SIMD)\n", ms * 1e6 / ((double)it2 * 4 * 8) * 2.4); }
    return 0;
}
