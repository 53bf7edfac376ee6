// Calibration of rocprofv3's FETCH_SIZE on gfx950 against a KNOWN byte count, for the two access widths the kernels of
// this library use: one dword per lane and 16 bytes per lane, streaming reads of a buffer larger than the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calib      (tools/profile_round.sh does this)
// Each kernel reads exactly 512 MiB once; FETCH_SIZE is reported in KB, so the ratio to 524288 is the factor.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read_dword(const float* __restrict__ x, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void calib_read_dwordx4(const float4* __restrict__ x, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    const size_t n = (size_t)128 << 20;  // 128 Mi floats = 512 MiB
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 4);
    hipMemset(x, 0, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        calib_read_dword<<<2048, 256>>>(x, n, out);
        calib_read_dwordx4<<<2048, 256>>>((const float4*)x, n / 4, out);
    }
    hipDeviceSynchronize();
    printf("read 512 MiB per launch, 3 launches per kernel\n");
    return 0;
}
