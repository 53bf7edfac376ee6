// fbsmi_sde.hip -- single-trajectory SDE paths and the linear-Gaussian model closures as
// stand-alone primitives for the closure-driven tier.
//
//   fbsmi_linear_path     : x_{k+1} = F_k x_k + S_k xi_k, the exact forward noising transition of
//                           make_linear_sde(sde).simulate_cond_forward, fbs/sdes/linear.py:190-221
//   fbsmi_affine_em_path  : Euler-Maruyama with sub-steps for a drift affine in x,
//                           x += (A_j x + B_j) ddt + S_j xi ; this is euler_maruyama
//                           (fbs/sdes/simulators.py:53-106) specialised to the Doob bridge drift of
//                           a scalar linear SDE (linear.py:36-45,83-92; simulators.py:126-160),
//                           i.e. bridge_sampler of fbs/samplers/gibbs.py:17-20
//   fbsmi_lg_transition_sampler / _likelihood_logpdf / _transition_logpdf :
//                           the closures of experiments/toy/gp_gibbs.py:120-135 on (n, du) row-major
//                           particles, same arithmetic as the fused sweep
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fbsmi.h"
#include "fbsmi_device.h"
#include "fbsmi_host.h"

namespace fbsmi {

// one thread per coordinate; xi is (T, D) row-major noise already drawn
__global__ void k_linear_path(const float* F, const float* S, const float* x0, const float* xi, int T, int64_t D,
                              float* out) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= D) return;
    float x = x0[c];
    out[c] = x;
    for (int k = 0; k < T; ++k) {
        x = F[k] * x + S[k] * xi[(int64_t)k * D + c];
        out[(int64_t)(k + 1) * D + c] = x;
    }
}

// keys (T,2) device; interval k draws normal(keys[k], (nsub, D)) (simulators.py:91); sub-step j of
// interval k uses coefficient row k*nsub + j.  B is (T*nsub) scalars multiplying `target`.
__global__ void k_affine_em_path(const uint32_t* keys, const float* A, const float* B, const float* S,
                                 const float* ddt, const float* target, const float* x0, int T, int nsub,
                                 int64_t D, int replace_last, float* out) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= D) return;
    float x = x0[c];
    const float tg = target[c];
    out[c] = x;
    const uint64_t n = (uint64_t)nsub * (uint64_t)D;
    for (int k = 0; k < T; ++k) {
        const uint32_t k0 = keys[2 * k], k1 = keys[2 * k + 1];
        const float h = ddt[k];
        const float sq = fbsmi_sqrtf(h);
        for (int j = 0; j < nsub; ++j) {
            const int r = k * nsub + j;
            const float xi = normal_at(k0, k1, n, (uint64_t)j * D + c);
            const float drift = A[r] * x + B[r] * tg;
            x = (x + drift * h) + (S[r] * sq) * xi;
        }
        out[(int64_t)(k + 1) * D + c] = x;
    }
    if (replace_last) out[(int64_t)T * D + c] = tg;
}

// One Euler-Maruyama sub-step for a drift the caller evaluated (any closure: a score network, an SDE's drift):
//   out = (x + drift * ddt) + c * xi,  xi = element `offset + e` of jax.random.normal(key, (n_total,)) drawn here
// -- the loop body of euler_maruyama, fbs/sdes/simulators.py:94-99, with c = dispersion(t) * sqrt(ddt); sub-step j of an
// interval is offset = j * x.size of the (integration_nsteps, *x.shape) draw of simulators.py:91.  Four elements per thread.
__global__ void __launch_bounds__(256) k_em_update(const float* __restrict__ x, const float* __restrict__ drift, float ddt,
                                                   float c, uint32_t k0, uint32_t k1, uint64_t n_total, uint64_t offset,
                                                   int64_t n, float* __restrict__ out) {
    const int64_t e0 = 4 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    if (e0 >= n) return;
    if (e0 + 4 <= n && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(drift) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const float4 xv = *reinterpret_cast<const float4*>(x + e0), dv = *reinterpret_cast<const float4*>(drift + e0);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (xs[i] + ds[i] * ddt) + c * normal_at(k0, k1, n_total, offset + (uint64_t)(e0 + i));
        *reinterpret_cast<float4*>(out + e0) = make_float4(o[0], o[1], o[2], o[3]);
        return;
    }
    for (int64_t e = e0; e < n && e < e0 + 4; ++e) out[e] = (x[e] + drift[e] * ddt) + c * normal_at(k0, k1, n_total, offset + (uint64_t)e);
}

// ---- LG closures on (n, du) row-major particles ---------------------------------------------
struct LgStep {
    const float* G;
    const float* g;
    float sd, sd2, lognorm, dt;
    int du, dv, D;
};

__device__ __forceinline__ float lg_drift(const LgStep& t, int r, const float* __restrict__ u,
                                          const float* __restrict__ v_prev) {
    const float* Gr = t.G + (size_t)r * t.D;
    float acc = t.g[r];
    for (int c = 0; c < t.du; ++c) acc = fbsmi_fmaf(Gr[c], u[c], acc);
    for (int c = 0; c < t.dv; ++c) acc = fbsmi_fmaf(Gr[t.du + c], v_prev[c], acc);
    return acc;
}

__device__ __forceinline__ float lg_norm_logpdf(float x, float loc, float sd2, float lognorm) {
    const float d = x - loc;
    return (lognorm + (d * d) / sd2) / -2.0f;
}

// rows [row0, row0 + n) of an ensemble of n_total rows: the noise index is the global one
__global__ void k_lg_transition_sampler(LgStep t, const float* us_prev, const float* v_prev, uint32_t k0,
                                        uint32_t k1, int64_t n_total, int64_t row0, int64_t n, float* us) {
    const int64_t total = n * t.du;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / t.du;
        const int r = (int)(e - p * t.du);
        const float* u = us_prev + p * t.du;
        const float dr = lg_drift(t, r, u, v_prev);
        const float xi = normal_at(k0, k1, (uint64_t)(n_total * t.du), (uint64_t)(row0 * t.du + e));
        us[e] = (u[r] + dr * t.dt) + t.sd * xi;
    }
}

// mode 0: likelihood_logpdf(v, us_prev, v_prev) (rows du..D-1, target v);
// mode 1: transition_logpdf(u, us_prev, v_prev) (rows 0..du-1, target u)
__global__ void k_lg_logpdf(LgStep t, int mode, const float* target, const float* us_prev, const float* v_prev,
                            int64_t n, float* lw) {
    const int rows = mode == 0 ? t.dv : t.du;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float* u = us_prev + p * t.du;
        float acc = 0.0f;
        for (int r = 0; r < rows; ++r) {
            const float dr = lg_drift(t, mode == 0 ? t.du + r : r, u, v_prev);
            const float mean = (mode == 0 ? v_prev[r] : u[r]) + dr * t.dt;
            const float lp = lg_norm_logpdf(target[r], mean, t.sd2, t.lognorm);
            acc = r == 0 ? lp : acc + lp;
        }
        lw[p] = acc;
    }
}

static inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace fbsmi

using namespace fbsmi;

#define FBSMI_NEED(cond, msg) \
    if (!(cond)) return fail(FBSMI_ERR_ARG, msg)
#define FBSMI_LAUNCH_CHECK()                                                     \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) return fail(FBSMI_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

// host copies of the per-step scalars are passed by the caller (they own the float64 -> float32
// tables); sd / lognorm are read on the host side of the ABI from HOST mirrors to build LgStep.
extern "C" {

int fbsmi_linear_path(const float* F, const float* S, const float* x0, const float* xi, int32_t T, int64_t D,
                      float* out, void* stream) {
    FBSMI_NEED(F && S && x0 && xi && out && T >= 0 && D >= 1, "linear_path: bad arguments");
    k_linear_path<<<grid_for(D, 64), 64, 0, (hipStream_t)stream>>>(F, S, x0, xi, T, D, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_affine_em_path(const uint32_t* keys, const float* A, const float* B, const float* S, const float* ddt,
                         const float* target, const float* x0, int32_t T, int32_t nsub, int64_t D, int replace_last,
                         float* out, void* stream) {
    FBSMI_NEED(keys && A && B && S && ddt && target && x0 && out && T >= 0 && nsub >= 1 && D >= 1,
               "affine_em_path: bad arguments");
    k_affine_em_path<<<grid_for(D, 64), 64, 0, (hipStream_t)stream>>>(keys, A, B, S, ddt, target, x0, T, nsub, D,
                                                                      replace_last, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_em_update(const float* x, const float* drift, float ddt, float c, uint32_t k0, uint32_t k1, int64_t n_total,
                    int64_t offset, int64_t n, float* out, void* stream) {
    FBSMI_NEED(n >= 0 && offset >= 0 && n_total >= offset + n && (n == 0 || (x && drift && out)), "em_update: bad arguments");
    if (n == 0) return FBSMI_OK;
    const int64_t blocks = ((n + 3) / 4 + 255) / 256;
    FBSMI_NEED(blocks <= 0x7fffffff, "em_update: too many elements");
    k_em_update<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, drift, ddt, c, k0, k1, (uint64_t)n_total,
                                                                             (uint64_t)offset, n, out);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

static int make_step(const fbsmi_lg_model* m, int32_t k, float sd, float lognorm, LgStep* t) {
    if (!m || k < 0 || k >= m->T) return fail(FBSMI_ERR_ARG, "lg closure: bad model / step index");
    const int D = m->du + m->dv;
    t->G = m->G + (size_t)k * D * D;
    t->g = m->g + (size_t)k * D;
    t->sd = sd;
    t->sd2 = sd * sd;
    t->lognorm = lognorm;
    t->dt = m->dt;
    t->du = m->du;
    t->dv = m->dv;
    t->D = D;
    return FBSMI_OK;
}

int fbsmi_lg_transition_sampler(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* us_prev,
                                const float* v_prev, uint32_t k0, uint32_t k1, int64_t n, float* us, void* stream) {
    LgStep t;
    int rc = make_step(m, k, sd_k, lognorm_k, &t);
    if (rc) return rc;
    FBSMI_NEED(us_prev && v_prev && us && n >= 0, "lg_transition_sampler: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_lg_transition_sampler<<<grid_for(n * t.du), 256, 0, (hipStream_t)stream>>>(t, us_prev, v_prev, k0, k1, n, 0, n, us);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_lg_transition_sampler_rows(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k,
                                     const float* us_prev, const float* v_prev, uint32_t k0, uint32_t k1,
                                     int64_t n_total, int64_t row0, int64_t n, float* us, void* stream) {
    LgStep t;
    int rc = make_step(m, k, sd_k, lognorm_k, &t);
    if (rc) return rc;
    FBSMI_NEED(us_prev && v_prev && us && n >= 0 && row0 >= 0 && row0 + n <= n_total,
               "lg_transition_sampler_rows: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_lg_transition_sampler<<<grid_for(n * t.du), 256, 0, (hipStream_t)stream>>>(t, us_prev, v_prev, k0, k1, n_total,
                                                                                 row0, n, us);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_lg_likelihood_logpdf(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* v,
                               const float* us_prev, const float* v_prev, int64_t n, float* lw, void* stream) {
    LgStep t;
    int rc = make_step(m, k, sd_k, lognorm_k, &t);
    if (rc) return rc;
    FBSMI_NEED(v && us_prev && v_prev && lw && n >= 0, "lg_likelihood_logpdf: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_lg_logpdf<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(t, 0, v, us_prev, v_prev, n, lw);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

int fbsmi_lg_transition_logpdf(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* u,
                               const float* us_prev, const float* v_prev, int64_t n, float* lw, void* stream) {
    LgStep t;
    int rc = make_step(m, k, sd_k, lognorm_k, &t);
    if (rc) return rc;
    FBSMI_NEED(u && us_prev && v_prev && lw && n >= 0, "lg_transition_logpdf: bad arguments");
    if (n == 0) return FBSMI_OK;
    k_lg_logpdf<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(t, 1, u, us_prev, v_prev, n, lw);
    FBSMI_LAUNCH_CHECK();
    return FBSMI_OK;
}

}  // extern "C"
