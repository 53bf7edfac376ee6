"""Gaussian-process regression with the twisted-SMC conditional sampler (Wu et al., 2023) on MI355X.

Counterpart of the reference driver experiments/toy/gp_twisted.py (same flags, key schedule and .npz schema:
samples (nsamples, d), gp_mean, gp_cov).  The twisting function's gradient goes through the score by torch
autograd, as the reference's does by jax.grad; resampling, gathers and normalisations are libfbsmi kernels."""
import argparse
import math
import os

import numpy as np
import torch

from _gp_toy import add_common_args, gp_setting
from fbs_amd import ops
from fbs_amd.samplers import stratified
from fbs_amd.samplers.smc import twisted_smc
from fbs_amd.sdes import make_linear_sde


def main(argv=None):
    args = add_common_args(argparse.ArgumentParser()).parse_args(argv)
    dev = torch.device('cuda:0')
    g = gp_setting(args, dev)
    key, ts, sde, d, N = g['key'], g['ts'], g['sde'], g['d'], args.nparticles
    T, dt, obs_var = float(ts[-1]), float(ts[1] - ts[0]), g['obs_var']
    y0 = g['y0_t']
    discretise = make_linear_sde(sde)[0]
    cov_mat = g['cov_mat']

    tabs = {}

    def score_tables(t):                                                            # gp_twisted.py:66-75
        """score(u, t) = -cov_t^{-1} (u - m_t) of the X-marginal (prior mean 0): the matrix -cov_t^{-1}."""
        k = round(float(t), 9)
        if k not in tabs:
            F_, Q_ = discretise(t, float(ts[0]))
            covt = float(F_) ** 2 * cov_mat + float(Q_) * np.eye(d)
            tabs[k] = torch.as_tensor(-np.linalg.inv(covt), dtype=torch.float32, device=dev)
        return tabs[k]

    def reverse_drift(u, t):                                                        # :83-84
        s_ = T - float(t)
        return -float(sde.drift(1.0, s_)) * u + float(sde.dispersion(s_)) ** 2 * (u @ score_tables(s_))

    def reverse_dispersion(t):
        return float(sde.dispersion(T - float(t)))

    def norm_logpdf_sum(x, loc, scale):
        return ((math.log(2 * math.pi * scale * scale) + (x - loc) ** 2 / (scale * scale)) / -2.0).sum(dim=-1)

    def twisting_logpdf(y, u, t):                                                   # :113-115
        return norm_logpdf_sum(y, u + reverse_drift(u, t) * dt, math.sqrt(obs_var))

    def reverse_cond_drift(u, t, y):                                                # :87-89 (jax.grad -> torch.autograd)
        with torch.enable_grad():
            uu = u.detach().requires_grad_(True)
            grad = torch.autograd.grad(twisting_logpdf(y, uu, t).sum(), uu)[0]
        return reverse_drift(u, t) + reverse_dispersion(t) ** 2 * grad

    def transition_logpdf(u, u_prev, t_prev):                                       # :100-104
        return norm_logpdf_sum(u, u_prev + reverse_drift(u_prev, t_prev) * dt, math.sqrt(dt) * reverse_dispersion(t_prev))

    F_T, Q_T = discretise(T, float(ts[0]))
    chol_ref = torch.as_tensor(np.linalg.cholesky(float(F_T) ** 2 * cov_mat + float(Q_T) * np.eye(d)).T.copy(),
                               dtype=torch.float32, device=dev)

    def init_sampler(key_, n_):                                                     # :107-110 (m_ref = 0)
        return ops.normal(key_, (n_, d), device=dev) @ chol_ref

    def twisting_prop_sampler(key_, us, t, y):                                      # :121-123
        m_ = us + reverse_cond_drift(us, t, y) * dt
        return m_ + math.sqrt(dt) * reverse_dispersion(t) * ops.normal(key_, (N, d), device=dev)

    def twisting_prop_logpdf(u, u_prev, t, y):                                      # :126-129
        m_ = u_prev + reverse_cond_drift(u_prev, t, y) * dt
        return norm_logpdf_sum(u, m_, math.sqrt(dt) * reverse_dispersion(t))

    def conditional_sampler(key_):                                                  # :133-141
        key_filter, key_select = ops.split(key_)
        uvs, log_ws = twisted_smc(key_filter, y0, ts, init_sampler, transition_logpdf, twisting_logpdf,
                                  twisting_prop_sampler, twisting_prop_logpdf, resampling=stratified, nparticles=N)
        return ops.choice(key_select, uvs, p=ops.math_map("exp", log_ws), axis=0)

    samples = torch.empty((args.nsamples, d), device=dev)
    for i in range(args.nsamples):                                                  # :144-148
        key, subkey = ops.split(key)
        samples[i] = conditional_sampler(subkey)
    samples = samples.cpu().numpy()
    if not args.quiet:
        err = np.abs(samples.mean(axis=0) - g['gp_mean']).max()
        print(f'ID: {args.id} | twisted | {args.nsamples} samples | max |mean - gp_mean| = {err:.3f}')
    os.makedirs(args.outdir, exist_ok=True)
    np.savez(os.path.join(args.outdir, f'twisted-{args.sde}-{args.nparticles}-{args.id}'),
             samples=samples, gp_mean=g['gp_mean'], gp_cov=g['gp_cov'])             # :151-152
    return samples, g['gp_mean'], g['gp_cov']


if __name__ == '__main__':
    main()
