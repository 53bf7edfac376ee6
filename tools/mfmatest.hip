// mfmatest -- operand layout and accumulation order of the f32 MFMA instructions on gfx950.
// For D = C + A(MxK) * B(KxN): which lane/register holds which element, and is
// D[i][j] == fma(a[i][K-1], b[K-1][j], ... fma(a[i][0], b[0][j], c[i][j])) bit for bit?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k16(const float* A, const float* B, const float* C, float* D, int K) {
    // 16x16x4: A[i][k] lane = i + 16*k (k<4); B[k][j] lane = j + 16*k; D[i][j]: lane = j + 16*(i/4), reg = i%4
    const int l = threadIdx.x;
    f4 acc;
    for (int v = 0; v < 4; ++v) acc[v] = C[(4 * (l / 16) + v) * 16 + (l % 16)];
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[(l % 16) * K + k0 + l / 16];
        const float b = B[(k0 + l / 16) * 16 + (l % 16)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + (l % 16)] = acc[v];
}
__global__ void k32(const float* A, const float* B, const float* C, float* D, int K) {
    // 32x32x2: A[i][k] lane = i + 32*k (k<2); B[k][j] lane = j + 32*k; D[i][j]: lane = j + 32*((i/4)%2), reg = (i%4) + 4*(i/8)
    const int l = threadIdx.x;
    f16v acc;
    for (int v = 0; v < 16; ++v) acc[v] = C[(8 * (v / 4) + 4 * (l / 32) + (v % 4)) * 32 + (l % 32)];
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[(l % 32) * K + k0 + l / 32];
        const float b = B[(k0 + l / 32) * 32 + (l % 32)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + (v % 4)) * 32 + (l % 32)] = acc[v];
}

static float frand(unsigned& s, bool wide) {
    s = s * 1664525u + 1013904223u;
    float x = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);
    if (wide) { s = s * 1664525u + 1013904223u; x *= ldexpf(1.0f, (int)(s >> 28) - 8); }
    return x;
}

int run(int M, int K, bool wide) {
    std::vector<float> A(M * K), B(K * M), C(M * M), D(M * M), R(M * M), R2(M * M), R3(M * M);
    unsigned s = 12345 + M + K + wide;
    for (auto& x : A) x = frand(s, wide);
    for (auto& x : B) x = frand(s, wide);
    for (auto& x : C) x = frand(s, wide);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    if (M == 16) k16<<<1, 64>>>(dA, dB, dC, dD, K); else k32<<<1, 64>>>(dA, dB, dC, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) {
        float acc = C[i * M + j], acc2 = C[i * M + j], acc3 = C[i * M + j];
        for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * M + j], acc);            // ascending fma chain
        for (int k = 0; k < K; ++k) { volatile float p = A[i * K + k] * B[k * M + j]; acc2 = acc2 + p; }  // unfused
        const int step = M == 16 ? 4 : 2;                                                       // descending inside an instruction
        for (int k0 = 0; k0 < K; k0 += step) for (int k = k0 + step - 1; k >= k0; --k) acc3 = fmaf(A[i * K + k], B[k * M + j], acc3);
        R[i * M + j] = acc; R2[i * M + j] = acc2; R3[i * M + j] = acc3;
    }
    int bad = 0, bad2 = 0, bad3 = 0; double maxrel = 0;
    for (int e = 0; e < M * M; ++e) {
        bad += memcmp(&D[e], &R[e], 4) != 0; bad2 += memcmp(&D[e], &R2[e], 4) != 0; bad3 += memcmp(&D[e], &R3[e], 4) != 0;
        maxrel = fmax(maxrel, fabs((double)D[e] - R[e]) / (fabs((double)R[e]) + 1e-30));
    }
    printf("%dx%dx%d K=%d %s: mismatches vs ascending fma chain %d, vs unfused %d, vs in-instruction descending %d of %d (max rel vs chain %.3g)\n",
           M, M, M == 16 ? 4 : 2, K, wide ? "wide-range" : "unit-range", bad, bad2, bad3, M * M, maxrel);
    return bad;
}
int main() {
    int bad = 0;
    for (int wide = 0; wide < 2; ++wide) { bad += run(16, 4, wide); bad += run(16, 200, wide); bad += run(32, 2, wide); bad += run(32, 200, wide); }
    printf(bad ? "LAYOUT OR ORDER DIFFERS\n" : "f32 MFMA == ascending fmaf chain, bit for bit, with the layouts in the source\n");
    return 0;
}
