// microbench5 -- what does a device-wide barrier inside one (cooperative) kernel cost on MI355X,
// against a kernel boundary?  Every workgroup publishes a value, all meet at the barrier, every
// workgroup reads another workgroup's value (checks cross-XCD visibility).  Spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
    bool ok = true;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(256) k_bar(unsigned* ctr, float* buf, int iters, int* bad) {
    const int nb = gridDim.x, b = blockIdx.x;
    for (int it = 1; it <= iters; ++it) {
        buf[(size_t)b * 256 + threadIdx.x] = (float)it;
        if (!grid_barrier(ctr, (unsigned)it * nb)) { if (threadIdx.x == 0) atomicAdd(bad, 1 << 16); return; }
        const int o = (b + nb / 2 + 1) % nb;
        if (buf[(size_t)o * 256 + threadIdx.x] != (float)it) atomicAdd(bad, 1);
        // second barrier so that nobody overwrites buf before it was read
        if (!grid_barrier(ctr + 64, (unsigned)it * nb)) { if (threadIdx.x == 0) atomicAdd(bad, 1 << 16); return; }
    }
}

__global__ void __launch_bounds__(256) k_step(float* buf, int it) {
    const int nb = gridDim.x, b = blockIdx.x;
    const int o = (b + nb / 2 + 1) % nb;
    const float v = buf[(size_t)o * 256 + threadIdx.x];
    buf[(size_t)(nb + b) * 256 + threadIdx.x] = v + (float)it;
}

int main(int argc, char** argv) {
    const int iters = 2000;
    unsigned* ctr; float* buf; int* bad;
    CK(hipMalloc(&ctr, 1024)); CK(hipMalloc(&buf, 2 * 4096 * 256 * 4)); CK(hipMalloc(&bad, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int maxb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_bar, 256, 0));
    printf("occupancy: %d blocks of 256 per CU\n", maxb);
    for (int nb : {64, 256, 512, 1024, 2048}) {
        CK(hipMemset(ctr, 0, 1024)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(buf, 0, 2 * 4096 * 256 * 4));
        int it = iters;
        void* args[] = {&ctr, &buf, &it, &bad};
        CK(hipEventRecord(e0, st));
        hipError_t e = hipLaunchCooperativeKernel((void*)k_bar, dim3(nb), dim3(256), args, 0, st);
        if (e != hipSuccess) { printf("grid %d: cooperative launch refused: %s\n", nb, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("grid %4d: %.3f us per barrier (write + barrier + remote read), bad=%d\n", nb, ms * 1e3 / (2.0 * iters), hb);
        // the same exchange as a chain of kernels in a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 500; ++i) k_step<<<nb, 256, 0, st>>>(buf, i);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %4d: %.3f us per kernel in a 500-node graph (read remote + write)\n", nb, ms * 1e3 / 500.0);
    }
    return 0;
}
