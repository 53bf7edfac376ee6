/*
 * fbsmi.h -- C ABI of libfbsmi, the MI355X (gfx950) engine for the particle-Gibbs / CSMC / pMCMC
 * hot path of zgbkdlm/fbs.
 *
 * The reference has no FFI: its boundary is the Python API of fbs.samplers / fbs.sdes on JAX
 * arrays.  Each entry point below names the reference function(s) whose
 * device work it performs; fbs_amd/ (Python, ctypes) keeps the reference's names and argument
 * orders on top of it, and INTEGRATION.md shows the binding a maintainer of the reference would
 * add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()) unless
 *     the parameter is documented "host";
 *   - `stream` is a hipStream_t (NULL = default stream); all work is stream-ordered, nothing
 *     synchronises, nothing allocates: scratch comes from the caller's workspace `ws`
 *     (fbsmi_workspace_bytes);
 *   - PRNG keys are JAX threefry keys, two uint32 passed by value as (k0, k1);
 *   - return value 0 = OK, negative = error (fbsmi_last_error() gives the text); no exceptions
 *     cross the ABI; calls are re-entrant across streams as long as workspaces differ;
 *   - float data is float32, indices int32, row-major.
 */
#ifndef FBSMI_H
#define FBSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBSMI_ABI_VERSION 1

#define FBSMI_OK 0
#define FBSMI_ERR_ARG (-1)
#define FBSMI_ERR_HIP (-2)
#define FBSMI_ERR_UNSUPPORTED (-3)

int fbsmi_abi_version(void);
const char* fbsmi_last_error(void);

/* ---- PRNG (jax.random.*; call sites listed in SURVEY.md Appendix A) ------------------------- */
/* jax.random.split(key, num) on the HOST (pure integer work; out is a host array of num*2). */
void fbsmi_key_split(uint32_t k0, uint32_t k1, int num, uint32_t* out_host);
int fbsmi_random_bits(uint32_t k0, uint32_t k1, int64_t n, uint32_t* out, void* stream);
int fbsmi_uniform(uint32_t k0, uint32_t k1, int64_t n, float* out, void* stream);
int fbsmi_normal(uint32_t k0, uint32_t k1, int64_t n, float* out, void* stream);
int fbsmi_randint(uint32_t k0, uint32_t k1, int64_t n, int32_t lo, int32_t hi, int32_t* out, void* stream);
/* Elements [start, start+count) of the flat draw of n_total elements (mode 0 bits, 1 uniform,
 * 2 normal): the slice a rank of a sharded particle ensemble owns, identical to the unsharded draw. */
int fbsmi_random_range(int mode, uint32_t k0, uint32_t k1, int64_t n_total, int64_t start, int64_t count, void* out,
                       void* stream);

/* ---- numeric specification probes (include/fbsmi_math.h evaluated on the device) ------------
 * op: 0 exp, 1 log, 2 log1p, 3 erfinv, 4 sqrt, 5 x/y, 6 bits->normal (x reinterpreted as uint32), 7 the same
 * through the kernels' branch-free device form (must equal 6 bit for bit on every input), 8 x/y through the kernels'
 * reciprocal-and-correction form (must equal 5 bit for bit). */
int fbsmi_math_map(int op, const float* x, const float* y, int64_t n, float* out, void* stream);

/* ---- tree reductions / scans -------------------------------------------------------------- */
size_t fbsmi_workspace_bytes(int64_t n);
/* jnp.cumsum (associative_scan order); out may alias x */
int fbsmi_cumsum(const float* x, int64_t n, float* out, void* ws, void* stream);
/* root of the pairwise tree over x (zero padded) -> out[0] */
int fbsmi_sum(const float* x, int64_t n, float* out, void* ws, void* stream);
/* jax.scipy.special.logsumexp -> out[0] */
int fbsmi_logsumexp(const float* x, int64_t n, float* out, void* ws, void* stream);
/* fbs/samplers/csmc/csmc.py:273-292 normalise: out = lw - logsumexp(lw) (exp'd unless log_space);
 * out may alias lw; out_lse (nullable) receives the logsumexp. */
int fbsmi_normalise(const float* lw, int64_t n, int log_space, float* out, float* out_lse, void* ws, void* stream);
/* The same with the two diagnostics a sharded / monitored run reports per step (SURVEY.md 8b `out_ess`; the reference
 * computes neither): out_lse (nullable) = logsumexp(lw), the increment of the log normalising constant when lw are the
 * unnormalised log-weights of a step (csmc.py:146, smc.py:145-146 `c`); out_ess (nullable) = 1 / sum_i w_i^2 with
 * w_i = exp(lw_i - lse) in float32, the sum being the root of the pairwise tree over the index bits. */
int fbsmi_normalise_ess(const float* lw, int64_t n, int log_space, float* out, float* out_lse, float* out_ess, void* ws,
                        void* stream);
/* jnp.searchsorted(a, q, side='left') for m queries */
int fbsmi_searchsorted(const float* a, int32_t n, const float* q, int64_t m, int32_t* out, void* stream);

/* ---- resamplers ---------------------------------------------------------------------------- */
/* fbs/samplers/resampling.py: kind 0 stratified :58, 1 systematic :54, 2 multinomial :62,
 * 3 killing :71.  Reference signature f(weights, key). */
int fbsmi_resample(int kind, const float* w, uint32_t k0, uint32_t k1, int32_t n, int32_t* idx, void* ws,
                   void* stream);
/* fbs/samplers/csmc/resamplings.py: kind 0 multinomial :10, 1 killing :40, 2 systematic :91
 * (conditional systematic is NotImplementedError in the reference -> FBSMI_ERR_UNSUPPORTED).
 * Reference signature f(key, weights, i, j, conditional). */
int fbsmi_cond_resample(int kind, uint32_t k0, uint32_t k1, const float* w, int32_t i, int32_t j, int conditional,
                        int32_t n, int32_t* idx, void* ws, void* stream);
/* jax.random.choice(key, n, (), p=w): csmc.py:295-297 barker_move, smc.py:104 -> out[0] */
int fbsmi_categorical(uint32_t k0, uint32_t k1, const float* w, int32_t n, int32_t* out, void* ws, void* stream);
/* fbs/samplers/gibbs.py:171-214 force_move -> out_i[0], out_alpha[0] (out_alpha nullable) */
int fbsmi_force_move(uint32_t k0, uint32_t k1, const float* w, int32_t k, int32_t n, int32_t* out_i,
                     float* out_alpha, void* ws, void* stream);

/* ---- data movement ------------------------------------------------------------------------- */
/* jnp.take(src, idx, axis=0): dst[r, :] = src[idx[r], :], rows of d floats (csmc.py:140) */
int fbsmi_gather_rows(const float* src, const int32_t* idx, int64_t n, int64_t d, float* dst, void* stream);
/* x.at[row].set(v): dst[row, :] = src[:]  (csmc.py:143,152) */
int fbsmi_set_row(float* dst, int64_t row, const float* src, int64_t d, void* stream);
/* csmc.py:262-267 ancestor back-trace: Bs[T] = B_T[0]; Bs[k-1] = As[k-1, Bs[k]]; As is (T, n) */
int fbsmi_backtrace(const int32_t* As, int32_t T, int32_t n, const int32_t* B_T, int32_t* Bs, void* stream);

/* ---- single-trajectory SDE paths ------------------------------------------------------------
 * out (T+1, D): out[0] = x0, out[k+1] = F[k]*out[k] + S[k]*xi[k]  -- the exact forward noising
 * transition of simulate_cond_forward(keep_path=True), fbs/sdes/linear.py:190-221; xi (T, D). */
int fbsmi_linear_path(const float* F, const float* S, const float* x0, const float* xi, int32_t T, int64_t D,
                      float* out, void* stream);
/* Euler-Maruyama with nsub sub-steps per interval for a drift affine in x:
 *   x += (A[r] x + B[r] target) ddt[k] + S[r] sqrt(ddt[k]) xi,  r = k*nsub + j,
 * xi = normal(keys[k], (nsub, D)) (fbs/sdes/simulators.py:53-106).  With the Doob bridge drift of a
 * scalar linear SDE this is doob_bridge_simulator (simulators.py:126-160), the bridge_sampler of
 * fbs/samplers/gibbs.py:17-20.  keys (T,2) uint32 device; out (T+1, D). */
int fbsmi_affine_em_path(const uint32_t* keys, const float* A, const float* B, const float* S, const float* ddt,
                         const float* target, const float* x0, int32_t T, int32_t nsub, int64_t D, int replace_last,
                         float* out, void* stream);

/* One Euler-Maruyama sub-step for a drift tensor the caller evaluated (a score network, any closure):
 *   out[e] = (x[e] + drift[e] * ddt) + c * xi[offset + e],  xi = jax.random.normal(key, (n_total,)) drawn in the kernel,
 * the loop body of euler_maruyama (fbs/sdes/simulators.py:94-99) with c = dispersion(t) * sqrt(ddt); sub-step j of an interval
 * is offset = j * n of the (integration_nsteps, *x.shape) draw of simulators.py:91.  out may alias x. */
int fbsmi_em_update(const float* x, const float* drift, float ddt, float c, uint32_t k0, uint32_t k1, int64_t n_total,
                    int64_t offset, int64_t n, float* out, void* stream);

/* ---- fused linear-Gaussian sampler ------------------------------------
 * The reverse drift of a scalar-coefficient linear SDE under a Gaussian prior is affine,
 * f(z, t_k) = G_k z + g_k; the three closures of experiments/toy/gp_gibbs.py:120-135 then need no
 * Python in the loop.  Tables (device, float32), T = number of steps, D = du + dv:
 *   G [T][D][D], g [T][D], sd [T] = sqrt(dt)*dispersion, lognorm [T] = log(2 pi sd^2),
 *   F [T], sqQ [T] forward transition ts[k] -> ts[k+1].                                        */
typedef struct fbsmi_lg_model {
    int32_t du, dv, T;
    float dt;
    const float* G;
    const float* g;
    const float* sd;
    const float* lognorm;
    const float* F;
    const float* sqQ;
} fbsmi_lg_model;

/* The three model closures on (n, du) ROW-MAJOR particles for step k (t_prev = ts[k]); sd_k and
 * lognorm_k are the host copies of sd[k], lognorm[k] (the tables themselves live on the device).
 * transition_sampler / likelihood_logpdf / transition_logpdf of experiments/toy/gp_gibbs.py:120-135. */
int fbsmi_lg_transition_sampler(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* us_prev,
                                const float* v_prev, uint32_t k0, uint32_t k1, int64_t n, float* us, void* stream);
/* transition_sampler for rows [row0, row0+n) of an ensemble of n_total rows (sharded ensembles):
 * us_prev / us hold only those n rows; the noise is the matching slice of the global draw. */
int fbsmi_lg_transition_sampler_rows(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k,
                                     const float* us_prev, const float* v_prev, uint32_t k0, uint32_t k1,
                                     int64_t n_total, int64_t row0, int64_t n, float* us, void* stream);
int fbsmi_lg_likelihood_logpdf(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* v,
                               const float* us_prev, const float* v_prev, int64_t n, float* lw, void* stream);
int fbsmi_lg_transition_logpdf(const fbsmi_lg_model* m, int32_t k, float sd_k, float lognorm_k, const float* u,
                               const float* us_prev, const float* v_prev, int64_t n, float* lw, void* stream);

typedef struct fbsmi_lg_sweep fbsmi_lg_sweep; /* opaque: device buffers + captured hipGraph */

/* Create the state for gibbs_kernel sweeps (fbs/samplers/gibbs.py:68-168, marg_y=False) with
 * `nparticles` particles, for `nchains` independent chains batched in every launch -- the
 * reference's jax.vmap over chains (experiments/toy/gp_gibbs.py:25,172-173).  store_path != 0 keeps
 * As / uss / log_wss (needed when explicit_backward == 0).  Allocates device memory (not
 * stream-ordered; call once).  All per-chain arrays below are laid out [nchains][...].
 * A handle owns one launch stream and one set of buffers: calls on the SAME handle must not overlap
 * (drive a handle from one host thread at a time); different handles are independent. */
int fbsmi_lg_sweep_create(const fbsmi_lg_model* model, int32_t nparticles, int explicit_backward,
                          int explicit_final, int store_path, int32_t nchains, fbsmi_lg_sweep** out);
void fbsmi_lg_sweep_destroy(fbsmi_lg_sweep* s);
/* One Gibbs sweep of every chain, everything on the device.  keys (C,2) uint32, x0 (C,du),
 * y0 (dv) shared, bs_star (C,T+1) are device inputs; x0_next (C,du), us_star_next (C,T+1,du),
 * bs_next (C,T+1), acc (C,T+1) bytes are device outputs (nullable; may alias the inputs of the next
 * call).  use_graph != 0 replays a hipGraph captured on the first call. */
int fbsmi_lg_gibbs_sweep(fbsmi_lg_sweep* s, const uint32_t* keys, const float* x0, const float* y0,
                         const int32_t* bs_star, float* x0_next, float* us_star_next, int32_t* bs_next,
                         uint8_t* acc, int use_graph, void* stream);
/* Chain `nsweeps` sweeps with the key schedule of the reference's drivers: per sweep
 * key, subkey = split(key); one chain sweeps with subkey (tests/test_gibbs.py:115-118), a batch of
 * C > 1 chains with split(subkey, C)[c] (experiments/toy/gp_gibbs.py:183-185).  key (2), x0 (C,du),
 * bs_star (C,T+1) are updated in place; x0s (nullable) receives (nsweeps, C, du). */
int fbsmi_lg_gibbs_chain(fbsmi_lg_sweep* s, uint32_t* key, float* x0, const float* y0, int32_t* bs_star,
                         int32_t nsweeps, float* x0s, int use_graph, void* stream);
/* A batch of chains may be driven as several handles ("groups") of fewer chains each, on their own streams: the step
 * kernels of a toy-sized ensemble are latency-bound (~3 us of every launch are its boundaries), so two half-size batches
 * whose launches interleave finish sooner than one full-size batch.  set_group tells a handle which chains of the batch it
 * drives (key schedule split(subkey, nchains_total)[first_chain + c], rows of x0s); call it before the handle's first
 * chain sweep.  chain_groups is fbsmi_lg_gibbs_chain over the handles (which must cover chains 0 .. nchains_total-1 in
 * order): same arguments, same results bit for bit, per-chain arrays laid out for the whole batch. */
int fbsmi_lg_sweep_set_group(fbsmi_lg_sweep* s, int32_t nchains_total, int32_t first_chain);
int fbsmi_lg_gibbs_chain_groups(fbsmi_lg_sweep* const* groups, int32_t ngroups, uint32_t* key, float* x0, const float* y0,
                                int32_t* bs_star, int32_t nsweeps, float* x0s, int use_graph, void* stream);
/* Parity views of the last sweep: copies view `which` into dst (device, nullable) and reports its
 * element count.  which: 0 final particles (n,du) row-major, 1 final normalised log-weights (n),
 * 2 As (T,n) int32, 3 uss (T+1,n,du), 4 log_wss (T+1,n) [2-4 only with store_path],
 * 5 us_star (T+1,du) and 6 vs (T+1,dv) of the sweep; each with a leading [nchains] axis;
 * 7 = 128 32-bit words of in-kernel clock stamps (only written by the -DFBSMI_STAMPS diagnostic build).
 * n = nparticles (+1 if explicit_final). */
int fbsmi_lg_sweep_view(fbsmi_lg_sweep* s, int which, void* dst, int64_t* count, void* stream);
/* Fused particle filters for the analytic model, stratified (resampling = 0) or systematic (1)
 * resampling (fbs/samplers/resampling.py:43-59):
 *   flow 0  bootstrap_filter (fbs/samplers/smc.py:9-88); loglik receives the NEGATIVE log-likelihood
 *           estimate; with store_path the filtering path (T+1, n, du) can be fetched (return_last=False);
 *   flow 1  pmcmc_filter_step (smc.py:115-158); loglik receives log_ell.
 * keys (C,2), vs (C,T+1,dv) the reversed observation path, u0s (C,n,du) row-major initial particles
 * are device inputs; uT (C,n,du), loglik (C), path (C,T+1,n,du) device outputs (nullable). */
typedef struct fbsmi_lg_filter fbsmi_lg_filter;
int fbsmi_lg_filter_create(const fbsmi_lg_model* model, int32_t nparticles, int flow, int resampling, int store_path,
                           int32_t nchains, fbsmi_lg_filter** out);
void fbsmi_lg_filter_destroy(fbsmi_lg_filter* f);
int fbsmi_lg_filter_run(fbsmi_lg_filter* f, const uint32_t* keys, const float* vs, const float* u0s, float* uT,
                        float* loglik, float* path, int use_graph, void* stream);

/* ---- fused SMC step for score-network models (image experiments) --------------------------------
 * The three closures of experiments/imgs/inpainting.py:102-147 (and supr.py; sb_imgs/supr.py:80-127)
 * wrap ONE network evaluation on the joint image concat(u, v) per SMC step (csmc.py:142,145 evaluate it
 * twice on the same input).  Around that evaluation (PyTorch-ROCm, not part of this library) the step is
 * two kernels:
 *   fbsmi_em_concat : ancestor gather (csmc.py:140) + ImageRestore.concat (fbs/data/images.py:355-363)
 *                     -> the network's input, written once in the network's dtype;
 *   fbsmi_em_finish : ImageRestore.unpack of the network output (images.py:333-353), reverse drift
 *                     (inpainting.py:102-103), Euler-Maruyama proposal with in-kernel
 *                     jax.random.normal (inpainting.py:122-128), reference pin (csmc.py:143) and the
 *                     row-summed Gaussian log-density of the observed increment (inpainting.py:141-147).
 * A particle row holds du unobserved floats (p, c) -> p*c + channel; an image holds D = du + dv floats
 * (w, h, c) row-major.  The mask is three int32 device tables:
 *   u_off (du): image offset of unobserved element j;  v_off (dv): image offset of observed element j;
 *   role (D): inverse map, role[e] = j >= 0 if image element e is unobserved element j, ~j < 0 if it is
 *   observed element j.                                                                              */
typedef struct fbsmi_em_mask {
    int32_t du, dv;
    const int32_t* u_off;
    const int32_t* v_off;
    const int32_t* role;
} fbsmi_em_mask;

/* img[r] = concat(us[A[r]], v_prev) for n rows; A nullable (identity).  out_dtype 0 float32, 1 bfloat16
 * (round to nearest even).  us (rows, du), v_prev (dv), img (n, D). */
int fbsmi_em_concat(const fbsmi_em_mask* mask, const float* us, const int32_t* A, const float* v_prev, int64_t n,
                    int out_dtype, void* img, void* stream);

/* net: the network evaluated on fbsmi_em_concat's output, (rows, D); net_dtype 0 float32, 1 bfloat16.  Row r of
 * this call uses net[net_A[r]] (net_A nullable: net[r]) -- pmcmc_filter_step (fbs/samplers/smc.py:144-150)
 * weights the particles, resamples, and proposes from the SAME network input rows, gathered.
 * mode 0: reverse drift = cx * x + cs * net (score model: cx = -a(T - t), cs = b(T - t)^2, inpainting.py:102-103);
 * mode 1: reverse drift = net (Schrodinger-bridge backward drift, sb_imgs/supr.py:84-85).
 * With x = us[A[r]] (A nullable), z = rows [row0, row0 + n) of jax.random.normal(key, (n_total, du)):
 *   us_new[r] = (x + drift_u * dt) + sd * z;   us_new[pin_row] = pin_value (pin_row < 0: no pin);
 *   lw[r] = tree-sum_j ( log(2 pi sd^2) + (v[j] - (v_prev[j] + drift_v[j] * dt))^2 / sd^2 ) / -2
 * (jax.scipy.stats.norm.logpdf summed in the canonical pairwise order of include/fbsmi_math.h).
 * us_new (n, du) nullable (no proposal), lw (n) nullable (no weights); us_new must not alias us. */
int fbsmi_em_finish(const fbsmi_em_mask* mask, const float* us, const int32_t* A, const void* net,
                    const int32_t* net_A, int net_dtype, int mode, float cx, float cs, float dt, float sd, const float* v, const float* v_prev, uint32_t k0,
                    uint32_t k1, int64_t n_total, int64_t row0, int64_t n, int64_t pin_row, const float* pin_value,
                    float* us_new, float* lw, void* stream);

/* transition_logpdf (inpainting.py:131-138): lw[r] = sum_j norm.logpdf(u[j]; us[r][j] + drift_u[r][j] * dt, sd)
 * for the n rows of us (n, du) and the network output net (n, D) on concat(us, v_prev); u (du). */
int fbsmi_em_transition_logpdf(const fbsmi_em_mask* mask, const float* us, const void* net, int net_dtype, int mode,
                               float cx, float cs, float dt, float sd, const float* u, int64_t n, float* lw,
                               void* stream);

/* HIP-event timing hooks: average duration in microseconds of the propagate ("Euler") kernel
 * over the launches since the last reset; 0 launches -> returns 0. Only measured when
 * fbsmi_lg_sweep_profile(s, 1) was set (events force non-graph launches). */
int fbsmi_lg_sweep_profile(fbsmi_lg_sweep* s, int enable);
int fbsmi_lg_sweep_kernel_us(fbsmi_lg_sweep* s, int which, double* avg_us, int64_t* launches);

#ifdef __cplusplus
}
#endif

#endif /* FBSMI_H */
