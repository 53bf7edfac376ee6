// occ_probe -- how many 256-thread workgroups with a given amount of dynamic LDS fit a CU (hipOccupancyMaxActiveBlocksPerMultiprocessor),
// and what a launch of such workgroups really does: `n` workgroups each spin ~20 us; the launch's duration tells the rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ float dyn[];
__global__ void __launch_bounds__(256) k_spin(float* out, long long ticks) {
    dyn[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = dyn[1];
}
int main(int argc, char** argv) {
    float* out; (void)hipMalloc(&out, 4096 * 4);
    (void)hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {52224, 53248, 53760, 54272, 54528, 55296, 65536, 81408, 81920}) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin, 256, lds);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best[3];
        int grids[3] = {512, 626, 768};
        for (int gi = 0; gi < 3; ++gi) {
            k_spin<<<grids[gi], 256, lds>>>(out, 2000);   // 20 us at 100 MHz
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            k_spin<<<grids[gi], 256, lds>>>(out, 2000);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&best[gi], a, b);
        }
        printf("dynamic LDS %6d B: occupancy API %d workgroups per CU; 20 us workgroups: grid 512 -> %.1f us, 626 -> %.1f us, 768 -> %.1f us\n",
               lds, nb, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3);
    }
    return 0;
}
