// stamps_wide -- in-kernel time stamps of the wide (matrix-core drift) step kernels, d = 100.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/fbsmi.h"
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 100, T = 200, du = 100, dv = 100, D = du + dv, C = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<float> G((size_t)T * D * D), g((size_t)T * D), sd(T), ln(T), F(T), sq(T);
    unsigned s = 1;
    for (auto& x : G) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 0.05f; }
    for (auto& x : g) { s = s * 1664525u + 1013904223u; x = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 0.1f; }
    for (int k = 0; k < T; ++k) { sd[k] = 0.07f; ln[k] = logf(6.2831853f * sd[k] * sd[k]); F[k] = 0.998f; sq[k] = 0.07f; }
    auto up = [](const std::vector<float>& v) { float* p; (void)hipMalloc(&p, v.size() * 4); (void)hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
    fbsmi_lg_model m{du, dv, T, 1.0f / T, up(G), up(g), up(sd), up(ln), up(F), up(sq)};
    fbsmi_lg_sweep* h = nullptr;
    if (fbsmi_lg_sweep_create(&m, N, 1, 0, 0, C, &h)) { printf("create failed: %s\n", fbsmi_last_error()); return 1; }
    uint32_t* key; float *x0, *y0; int32_t* bs;
    (void)hipMalloc(&key, 8); (void)hipMalloc(&x0, 4 * C * du); (void)hipMalloc(&y0, 4 * dv); (void)hipMalloc(&bs, (T + 1) * 4 * C);
    (void)hipMemset(key, 1, 8); (void)hipMemset(x0, 0, 4 * C * du); (void)hipMemset(y0, 0, 4 * dv); (void)hipMemset(bs, 0, (T + 1) * 4 * C);
    hipStream_t st; (void)hipStreamCreate(&st);
    fbsmi_lg_gibbs_chain(h, key, x0, y0, bs, 5, nullptr, 1, st); (void)hipStreamSynchronize(st);
    unsigned long long* d; (void)hipMalloc(&d, 64 * 8); int64_t cnt = 0;
    fbsmi_lg_sweep_view(h, 7, d, &cnt, st); (void)hipStreamSynchronize(st);
    unsigned long long hh[64]; (void)hipMemcpy(hh, d, sizeof(hh), hipMemcpyDeviceToHost);
    auto rt = [&](int i) { return (double)hh[2 * i] * 10.0; };   // ns
    const char* names[32] = {};
    names[25] = "G tile + noise under way, pre in"; names[26] = "pre row sums done"; names[27] = "pre lse known"; names[28] = "pre CDFs in LDS"; names[29] = "pre out";
    names[20] = "gemm in"; names[21] = "gemm ancestor rows landed"; names[22] = "gemm tiles in LDS"; names[23] = "gemm MFMA done"; names[24] = "gemm out";
    if (N > 256) {   // k_lgw_gemm_fat: entry, tiles staged, tile 0 (MFMA done, epilogue done, G swapped), tile 5 likewise, exit
        const char* fn[32] = {};
        fn[20] = "fat in"; fn[21] = "fat G tile 0 + ancestor rows in LDS"; fn[22] = "fat tile 0 MFMA done (noise drawn before)";
        fn[23] = "fat tile 0 epilogue done"; fn[24] = "fat tile 1 in LDS"; fn[25] = "fat tile 5 in LDS"; fn[26] = "fat tile 5 MFMA done";
        fn[27] = "fat tile 5 epilogue done"; fn[30] = "fat out";
        int fo[] = {20, 21, 22, 23, 24, 25, 26, 27, 30};
        double f0 = rt(20), fp = f0;
        for (int i : fo) { printf("%-44s t=%8.0f ns  (+%6.0f)\n", fn[i], rt(i) - f0, rt(i) - fp); fp = rt(i); }
        {   // every workgroup's entry / exit (view 9)
            const int nwg = ((N + 31) / 32) * C;
            unsigned long long* dv; (void)hipMalloc(&dv, 16384 * 4);
            fbsmi_lg_sweep_view(h, 9, dv, &cnt, st); (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> sp(8192);
            (void)hipMemcpy(sp.data(), dv, 8192 * 8, hipMemcpyDeviceToHost);
            if (nwg <= 4096) {
                unsigned long long t0 = ~0ull, t1 = 0; double life = 0, lmax = 0, lmin = 1e18;
                for (int w = 0; w < nwg; ++w) { t0 = sp[2 * w] < t0 ? sp[2 * w] : t0; t1 = sp[2 * w + 1] > t1 ? sp[2 * w + 1] : t1; }
                int late = 0; unsigned long long lastin = 0;
                for (int w = 0; w < nwg; ++w) {
                    const double l = (double)(sp[2 * w + 1] - sp[2 * w]) * 10.0;
                    life += l; lmax = l > lmax ? l : lmax; lmin = l < lmin ? l : lmin;
                    if ((sp[2 * w] - t0) * 10 > 5000) ++late;
                    lastin = sp[2 * w] > lastin ? sp[2 * w] : lastin;
                }
                printf("%d workgroups: first entry -> last exit %.0f ns; lifetimes %.0f .. %.0f ns (mean %.0f); last entry at +%.0f ns; %d entered more than 5 us after the first\n",
                       nwg, (double)(t1 - t0) * 10.0, lmin, lmax, life / nwg, (double)(lastin - t0) * 10.0, late);
            }
        }
        // shader-clock counter against the 100 MHz wall clock over the workgroup's lifetime: the clock the kernel really ran at
        const double dsh = (double)(hh[2 * 30 + 1] - hh[2 * 20 + 1]), dwall = rt(30) - rt(20);
        printf("(d=100, N=%d, chains=%d; the last step of a sweep; s_memtime / wall = %.3f ticks per ns)\n", N, C, dsh / dwall);
        return 0;
    }
    names[31] = "pre own draws done (row loads in flight)";
    names[30] = "gemm outputs stored";
    names[24] = "gemm out (next step's noise share drawn)";
    int order[] = {20, 25, 31, 26, 27, 28, 29, 21, 22, 23, 30, 24};
    double t0 = rt(20), prev = t0;
    for (int i : order) { printf("%-32s t=%8.0f ns  (+%6.0f)\n", names[i], rt(i) - t0, rt(i) - prev); prev = rt(i); }
    printf("(d=100, N=%d, chains=%d; the last step of a sweep)\n", N, C);
    return 0;
}
