// boundary_probe -- what a dependent kernel boundary inside a hipGraph is made of on MI355X: every kernel of a chain of
// small kernels (grid x 256 threads) records, in its block 0 and in its last block,
//   t0: s_memrealtime issued as the kernel's first instruction (before any kernel argument is needed),
//   t1: after the kernel-argument segment has arrived (first use of a pointer argument),
//   t2: after a global load of data the previous kernel wrote,
//   t3: at its end (after its store has been issued).
// 100 MHz wall clock (10 ns ticks).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/boundary_probe tools/boundary_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { float* buf[2]; unsigned long long* log; int pad[130]; };   // ~560 bytes by value, like LgDev

__global__ void __launch_bounds__(256) k_step(Big b, int s) {
    unsigned long long t0, t1, t2, t3;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const float* in = b.buf[s & 1];
    float* out = b.buf[(s & 1) ^ 1];
    asm volatile("s_nop 0" ::"s"(in), "s"(out) : "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const int i = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    const float v = in[(i + 4099) % n];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    out[i] = v + 1.0f;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        unsigned long long* l = b.log + ((size_t)s * 2 + (blockIdx.x ? 1 : 0)) * 4;
        l[0] = t0; l[1] = t1; l[2] = t2; l[3] = t3;
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int K = 400;
    for (int grid : {64, 256, 512}) {
        Big b{};
        CK(hipMalloc(&b.buf[0], grid * 256 * 4)); CK(hipMalloc(&b.buf[1], grid * 256 * 4));
        CK(hipMemset(b.buf[0], 0, grid * 256 * 4)); CK(hipMemset(b.buf[1], 0, grid * 256 * 4));
        CK(hipMalloc(&b.log, sizeof(unsigned long long) * K * 8));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int s = 0; s < K; ++s) k_step<<<grid, 256, 0, st>>>(b, s);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(K * 8);
        CK(hipMemcpy(h.data(), b.log, sizeof(unsigned long long) * K * 8, hipMemcpyDeviceToHost));
        double a01 = 0, a12 = 0, a23 = 0, gap0 = 0, gapl = 0, period = 0, spread = 0;
        int n = 0;
        for (int s = 100; s < K - 1; ++s, ++n) {
            const unsigned long long* f = &h[(size_t)s * 8];        // first block
            const unsigned long long* l = &h[(size_t)s * 8 + 4];    // last block
            const unsigned long long* nf = &h[(size_t)(s + 1) * 8];
            a01 += (double)(f[1] - f[0]); a12 += (double)(f[2] - f[1]); a23 += (double)(f[3] - f[2]);
            const unsigned long long end = l[3] > f[3] ? l[3] : f[3];
            gap0 += (double)(nf[0] - end);                          // last stamp of kernel s -> first instruction of kernel s + 1
            spread += (double)((long long)l[0] - (long long)f[0]);  // first block's start -> last block's start
            period += (double)(nf[0] - f[0]);
        }
        printf("grid %4d x 256: period %.2f us per kernel = start->kernarg %.2f + kernarg->loaded %.2f + ->stored %.2f + end->next start %.2f; "
               "first->last block start %.2f us\n", grid, period / n * 0.01, a01 / n * 0.01, a12 / n * 0.01, a23 / n * 0.01, gap0 / n * 0.01,
               spread / n * 0.01);
        (void)gapl;
    }
    return 0;
}
