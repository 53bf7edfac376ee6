"""Gaussian-process regression with the exact conditional score (Song et al., 2021) on MI355X.

Counterpart of the reference driver experiments/toy/gp_csgm.py (same flags, key schedule and .npz schema `csgm-<sde>-<id>.npz`:
samples (nsamples, d), gp_mean, gp_cov) -- the `csgm` column of the paper's Table 1 (experiments/tabulators/tabulate_toy.py:22).
The reverse drift is -a u + b^2 (grad log p_t(u) + grad_u log p(y0 | u_t = u)); both scores are Gaussian, so the drift is
affine in u: the reference gets the second term by jax.grad of a multivariate-normal log-density, here the same gradient is
written out (cond_m is affine in u: grad = M^T cond_cov^{-1} (y0 - cond_m(u))) and tabulated per step in float64.  The
integration is fbs_amd's euler_maruyama (one libfbsmi kernel per step with the noise drawn inside, fbsmi_em_update)."""
import argparse
import os

import numpy as np
import torch

from _gp_toy import add_common_args, gp_setting
from fbs_amd import ops
from fbs_amd.sdes import make_linear_sde
from fbs_amd.sdes.simulators import euler_maruyama


def main(argv=None):
    p = add_common_args(argparse.ArgumentParser())
    p.set_defaults(d=100)                                                           # gp_csgm.py:12
    args = p.parse_args(argv)
    dev = torch.device('cuda:0')
    g = gp_setting(args, dev)
    key, ts, sde, d = g['key'], g['ts'], g['sde'], g['d']
    T, cov_mat, obs_var, y0 = float(ts[-1]), g['cov_mat'], g['obs_var'], g['y0'].astype(np.float64)
    discretise = make_linear_sde(sde)[0]
    eye = np.eye(d)
    Kyy = cov_mat + obs_var * eye

    # terminal reference distribution (gp_csgm.py:70-78; the reference multiplies the noise by the covariance itself)
    F_ref, Q_ref = (float(x) for x in discretise(T, float(ts[0])))
    cond_m_ref = F_ref * cov_mat @ np.linalg.solve(Kyy, y0)
    cond_cov_ref = F_ref ** 2 * cov_mat + Q_ref * eye - F_ref * cov_mat @ np.linalg.solve(Kyy, F_ref * cov_mat)
    m_ref_t = torch.as_tensor(cond_m_ref, dtype=torch.float32, device=dev)
    cov_ref_t = torch.as_tensor(cond_cov_ref, dtype=torch.float32, device=dev)

    tabs = {}

    def drift_tables(t):
        """reverse_drift(u, t) = u A^T + c at reverse time t (gp_csgm.py:81-94), float64 on the host."""
        k = round(float(t), 9)
        if k not in tabs:
            s_ = T - float(t)
            F, Q = (float(x) for x in discretise(s_, float(ts[0])))
            Sx = F ** 2 * cov_mat + Q * eye
            Sx_inv = np.linalg.inv(Sx)
            M = F * cov_mat @ Sx_inv                                   # cond_m = M x_
            cond_cov = Kyy - M @ (F * cov_mat)
            Gm = M.T @ np.linalg.inv(cond_cov)                         # grad_u logpdf(y0; M u, cond_cov) = Gm (y0 - M u)
            a, b2 = float(sde.drift(1.0, s_)), float(sde.dispersion(s_)) ** 2
            A = -a * eye + b2 * (-Sx_inv - Gm @ M)
            c = b2 * (Gm @ y0)
            tabs[k] = (torch.as_tensor(A.T.copy(), dtype=torch.float32, device=dev),
                       torch.as_tensor(c, dtype=torch.float32, device=dev))
        return tabs[k]

    def reverse_drift(u, t):
        At, c = drift_tables(t)
        return u @ At + c

    def reverse_dispersion(t):
        return float(sde.dispersion(T - float(t)))

    def conditional_sampler(key_):                                                  # gp_csgm.py:103-108
        key_init, key_sde = ops.split(key_, 2)
        u0 = m_ref_t + cov_ref_t @ ops.normal(key_init, (d,), device=dev)
        return euler_maruyama(key_sde, u0, ts, reverse_drift, reverse_dispersion, integration_nsteps=1, return_path=False)

    samples = torch.empty((args.nsamples, d), device=dev)
    for i in range(args.nsamples):                                                  # gp_csgm.py:111-116
        key, subkey = ops.split(key)
        samples[i] = conditional_sampler(subkey)
    samples = samples.cpu().numpy()
    if not args.quiet:
        err = np.abs(samples.mean(axis=0) - g['gp_mean']).max()
        print(f'ID: {args.id} | csgm | {args.nsamples} samples | max |mean - gp_mean| = {err:.3f}')
    os.makedirs(args.outdir, exist_ok=True)
    np.savez(os.path.join(args.outdir, f'csgm-{args.sde}-{args.id}'), samples=samples, gp_mean=g['gp_mean'], gp_cov=g['gp_cov'])
    return samples, g['gp_mean'], g['gp_cov']


if __name__ == '__main__':
    main()
