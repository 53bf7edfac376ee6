// stamps -- run sweeps on the -DFBSMI_STAMPS diagnostic build of libfbsmi and print where one
// workgroup of each step kernel spends its time (100 MHz wall clock; shader clock for the GHz).
#include <hip/hip_runtime.h>
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
    fbsmi_lg_sweep* s = nullptr;
    if (fbsmi_lg_sweep_create(&m, N, 1, 0, 0, C, &s)) { printf("create failed: %s\n", fbsmi_last_error()); return 1; }
    uint32_t* key; float *x0, *y0; int32_t* bs;
    (void)hipMalloc(&key, 8); (void)hipMalloc(&x0, 4 * C); (void)hipMalloc(&y0, 4); (void)hipMalloc(&bs, (T + 1) * 4 * C);
    (void)hipMemset(key, 1, 8); (void)hipMemset(x0, 0, 4 * C); (void)hipMemset(y0, 0, 4); (void)hipMemset(bs, 0, (T + 1) * 4 * C);
    hipStream_t st; (void)hipStreamCreate(&st);
    fbsmi_lg_gibbs_chain(s, key, x0, y0, bs, 5, nullptr, 1, st); (void)hipStreamSynchronize(st);
    unsigned long long* d; (void)hipMalloc(&d, 64 * 8); int64_t cnt = 0;
    fbsmi_lg_sweep_view(s, 7, d, &cnt, st); (void)hipStreamSynchronize(st);
    unsigned long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    auto rt = [&](int i) { return (double)h[2 * i] * 10.0; };   // ns
    const char* names[] = {"", "", "norm in", "norm out", "cdf in", "cdf out", "prop in", "prop r0 issued+ALU", "prop heap barrier",
                           "prop J found", "prop w[src], LDS levels", "prop K rounds", "prop gather+compute+store", "prop out",
                           "norm lse known", "cdf phase 1 (loads, exchange)", "cdf tile (P,E) known"};
    // the last step of the loop: norm(2,14,3) cdf(4,15,16,5) prop(6..13)
    double t0 = rt(2);
    int order[] = {2, 14, 3, 4, 15, 16, 5, 6, 7, 8, 9, 10, 11, 12, 13};
    double prev = t0;
    for (int i : order) { printf("%-22s t=%8.0f ns  (+%6.0f)\n", names[i], rt(i) - t0, rt(i) - prev); prev = rt(i); }
    {   // launch spans of the last step: per-workgroup entry / exit stamps (FBSMI_SPAN_IN / _OUT, view 8)
        int64_t n8 = 0;
        fbsmi_lg_sweep_view(s, 8, nullptr, &n8, st);
        float* d8; (void)hipMalloc(&d8, n8 * 4);
        fbsmi_lg_sweep_view(s, 8, d8, &n8, st); (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> w(n8 / 2);
        (void)hipMemcpy(w.data(), d8, (n8 / 2) * 8, hipMemcpyDeviceToHost);
        // chain 0's array: norm stamps at [0, 1024), prop stamps at [1024, 2048) (u64 units); grids: norm 256 x C blocks of
        // 256 threads, prop (256 / halves) x C
        auto span = [&](int base, int blocks, const char* name, double& first_in, double& last_out) {
            double mn_in = 1e300, mx_in = 0, mn_out = 1e300, mx_out = 0, sum = 0; int cnt = 0;
            for (int b = 0; b < blocks; ++b) {
                const double in = (double)w[base + 2 * b] * 10.0, out = (double)w[base + 2 * b + 1] * 10.0;
                if (in == 0 || out == 0) continue;
                mn_in = in < mn_in ? in : mn_in; mx_in = in > mx_in ? in : mx_in;
                mn_out = out < mn_out ? out : mn_out; mx_out = out > mx_out ? out : mx_out; sum += out - in; ++cnt;
            }
            printf("%-5s %4d workgroups: first in -> last out %6.0f ns; entries spread over %5.0f ns, exits over %5.0f ns; mean in -> out %6.0f ns\n",
                   name, cnt, mx_out - mn_in, mx_in - mn_in, mx_out - mn_out, sum / (cnt ? cnt : 1));
            first_in = mn_in; last_out = mx_out;
        };
        double ni, no, pi, po;
        span(0, 512, "norm", ni, no);
        span(1024, 512, "prop", pi, po);
        printf("last norm workgroup out -> first prop workgroup in: %6.0f ns\n", pi - no);
    }
    double dclk = (double)(h[2 * 13 + 1] - h[2 * 6 + 1]), dns = rt(13) - rt(6);
    printf("shader clock during prop: %.2f GHz (N=%d chains=%d)\n", dclk / dns, N, C);
    return 0;
}
