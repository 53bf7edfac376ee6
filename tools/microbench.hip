// microbench.hip -- measures the fixed costs that bound a latency-limited SMC step on MI355X:
// kernel-to-kernel boundary in a hipGraph, dependent global-memory round trips, workgroup barriers.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_load1(const float* a, float* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i] + 1.0f; }
// chain of `depth` dependent loads through an index array (pointer chase), then a store
__global__ void k_chase(const int* nxt, const float* a, float* b, int n, int depth) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    int j = i; for (int d = 0; d < depth; ++d) j = nxt[j];
    b[i] = a[j] + 1.0f;
}
__global__ void k_sync(const float* a, float* b, int n, int nsync) {
    __shared__ float s[4]; int i = blockIdx.x * blockDim.x + threadIdx.x; float v = i < n ? a[i] : 0.f;
    for (int k = 0; k < nsync; ++k) { if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v; __syncthreads(); v += s[(k + 1) & 3]; __syncthreads(); }
    if (i < n) b[i] = v;
}
__global__ void k_shfl(const float* a, float* b, int n, int nshfl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; float v = i < n ? a[i] : 0.f;
    for (int k = 0; k < nshfl; ++k) v += __shfl_xor(v, 1 << (k % 6));
    if (i < n) b[i] = v;
}

template <typename F> double time_graph(hipStream_t st, int reps, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int r = 0; r < reps; ++r) enqueue(r);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); for (int it = 0; it < 5; ++it) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.0 / (5.0 * reps);
}

int main() {
    const int n = 65536; hipStream_t st; CK(hipStreamCreate(&st));
    float *a, *b; int* nxt; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&nxt, n * 4));
    std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 40503u + 12345u) % n;
    CK(hipMemcpy(nxt, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    const int reps = 1000;
    struct Cfg { int grid, block; } cfgs[] = {{256, 256}, {64, 256}, {64, 1024}, {16, 1024}, {1024, 64}};
    for (auto c : cfgs) {
        double t0 = time_graph(st, reps, [&](int) { k_empty<<<c.grid, c.block, 0, st>>>(); });
        double t1 = time_graph(st, reps, [&](int r) { (r & 1) ? k_load1<<<c.grid, c.block, 0, st>>>(b, a, n) : k_load1<<<c.grid, c.block, 0, st>>>(a, b, n); });
        printf("grid %4d x %4d : empty %.2f us/kernel, load+store (ping-pong) %.2f\n", c.grid, c.block, t0, t1);
        for (int depth : {1, 2, 4, 8, 17}) {
            double t = time_graph(st, reps, [&](int r) { (r & 1) ? k_chase<<<c.grid, c.block, 0, st>>>(nxt, b, a, n, depth) : k_chase<<<c.grid, c.block, 0, st>>>(nxt, a, b, n, depth); });
            printf("    chase depth %2d : %.2f us/kernel\n", depth, t);
        }
        for (int ns : {1, 4, 8}) {
            double t = time_graph(st, reps, [&](int r) { (r & 1) ? k_sync<<<c.grid, c.block, 0, st>>>(b, a, n, ns) : k_sync<<<c.grid, c.block, 0, st>>>(a, b, n, ns); });
            printf("    %d x (lds write+2 barriers) : %.2f us/kernel\n", ns, t);
        }
        double ts = time_graph(st, reps, [&](int r) { (r & 1) ? k_shfl<<<c.grid, c.block, 0, st>>>(b, a, n, 12) : k_shfl<<<c.grid, c.block, 0, st>>>(a, b, n, 12); });
        printf("    12 shuffles : %.2f us/kernel\n", ts);
    }
    // eager (non-graph) launch rate for comparison
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 100; ++i) k_empty<<<256, 256, 0, st>>>();
        hipStreamSynchronize(st); hipEventRecord(e0, st);
        for (int i = 0; i < 2000; ++i) k_load1<<<256, 256, 0, st>>>((i & 1) ? b : a, (i & 1) ? a : b, n);
        hipEventRecord(e1, st); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("eager stream launches, load+store 256x256: %.2f us/kernel\n", ms * 1000 / 2000);
    }
    return 0;
}
