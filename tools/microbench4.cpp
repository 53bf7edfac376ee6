// microbench4 -- drive libfbsmi's fused sweep from a bare C++ host (no Python / torch in the process).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/fbsmi.h"
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 500, D = 2, C = argc > 2 ? atoi(argv[2]) : 1;
    std::vector<float> G(T * D * D), g(T * D), sd(T), ln(T), F(T), sq(T);
    for (int k = 0; k < T; ++k) { G[k*4] = -0.3f; G[k*4+1] = 0.1f; G[k*4+2] = 0.1f; G[k*4+3] = -0.4f; g[k*2] = 0.1f; g[k*2+1] = -0.1f;
        sd[k] = 0.0632f; ln[k] = logf(6.2831853f * sd[k] * sd[k]); F[k] = 0.998f; sq[k] = 0.0632f; }
    auto up = [](const std::vector<float>& v) { float* p; (void)hipMalloc(&p, v.size() * 4); (void)hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
    fbsmi_lg_model m{1, 1, T, 2.0f / T, up(G), up(g), up(sd), up(ln), up(F), up(sq)};
    const int H = argc > 3 ? atoi(argv[3]) : 1;   // independent handles driven concurrently, each on its own stream
    std::vector<fbsmi_lg_sweep*> hs(H, nullptr);
    std::vector<hipStream_t> sts(H);
    std::vector<uint32_t*> keys(H);
    float *x0, *y0; int32_t* bs;
    (void)hipMalloc(&x0, 4 * C); (void)hipMalloc(&y0, 4); (void)hipMalloc(&bs, (T + 1) * 4 * C);
    (void)hipMemset(x0, 0, 4 * C); (void)hipMemset(y0, 0, 4); (void)hipMemset(bs, 0, (T + 1) * 4 * C);
    for (int h = 0; h < H; ++h) {
        if (fbsmi_lg_sweep_create(&m, N, 1, 0, 0, C, &hs[h])) { printf("create failed: %s\n", fbsmi_last_error()); return 1; }
        (void)hipStreamCreateWithFlags(&sts[h], hipStreamNonBlocking);
        (void)hipMalloc(&keys[h], 8); (void)hipMemset(keys[h], 1 + h, 8);
    }
    (void)hipDeviceSynchronize();
    for (int h = 0; h < H; ++h) fbsmi_lg_gibbs_chain(hs[h], keys[h], x0, y0, bs, 2, nullptr, 1, sts[h]);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 10; ++it)
        for (int h = 0; h < H; ++h) fbsmi_lg_gibbs_chain(hs[h], keys[h], x0, y0, bs, 1, nullptr, 1, sts[h]);
    (void)hipDeviceSynchronize();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 10;
    printf("bare C++ host, N=%d, chains/handle=%d, concurrent handles=%d: %.3f ms/sweep = %.2f us/step = %.3f G particle-steps/s\n", N, C, H, dt * 1e3, dt / T * 1e6, (double)N * T * C * H / dt / 1e9);
    for (auto h : hs) fbsmi_lg_sweep_destroy(h);
    return 0;
}
