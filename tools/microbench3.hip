// microbench3.hip -- the library's own step kernels launched from a bare harness (same code, no
// sweep object), to separate kernel cost from graph/stream effects.
#include "../fbs_amd/csrc/fbsmi_prims.hip"
#include "../fbs_amd/csrc/fbsmi_lg.hip"
#include <cstdio>
template <typename F> double time_graph(hipStream_t st, int reps, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    for (int r = 0; r < reps; ++r) enqueue(r);
    (void)hipStreamEndCapture(st, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st); for (int it = 0; it < 5; ++it) (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms * 1000.0 / (5.0 * reps);
}
int main() {
    const int N = 65536, nb = 256; 
    LgDev d{}; d.N = N; d.nparticles = N; d.du = 1; d.dv = 1; d.D = 2; d.T = 500; d.nb = nb; d.levels = bisect_levels(N);
    auto A = [](size_t n) { void* p; (void)hipMalloc(&p, n * 4); (void)hipMemset(p, 0, n * 4); return (float*)p; };
    d.lw = A(N); d.bmax = A(nb); d.bsumexp = A(nb); d.bsumw = A(nb); d.bsumJ = A(nb); d.w = A(N); d.lwn = A(N); d.scal = A(16);
    d.bs = (int32_t*)A(501); d.cdf = A(N); d.cdfJ = A(N);
    for (int mode = 0; mode < 2; ++mode) {
        hipStream_t st; if (mode) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); else (void)hipStreamCreate(&st);
        printf("stream %s: sumexp %.2f us, norm %.2f us\n", mode ? "nonblocking" : "default-created",
               time_graph(st, 1000, [&](int) { k_lg_sumexp<1><<<nb, 256, 0, st>>>(d); }),
               time_graph(st, 1000, [&](int r) { k_lg_norm<1, 0><<<nb, 256, 0, st>>>(d, r % 500); }));
    }
    return 0;
}
