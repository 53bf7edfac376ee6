"""Gaussian-process regression with the forward-backward Gibbs sampler on MI355X.

Counterpart of the reference driver experiments/toy/gp_gibbs.py (same command-line flags, same
key schedule, same .npz schema: samples (nchains, nsamples, d), gp_mean, gp_cov), written against
fbs_amd.  The analytic score makes the model a LinearGaussianBridge, so every sweep of all chains is
one hipGraph replay on the device.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd  # noqa: E402
from fbs_amd import ops  # noqa: E402
from fbs_amd.samplers import bootstrap_filter, stratified  # noqa: E402
from fbs_amd.samplers.smc import bootstrap_backward_smoother  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--d', type=int, default=10, help='The problem dimension.')
    parser.add_argument('--nparticles', type=int, default=10, help='The number of particles.')
    parser.add_argument('--nsamples', type=int, default=1000, help='The number of samples to draw.')
    parser.add_argument('--sde', type=str, default='const', help='The type of forward SDE.')
    parser.add_argument('--explicit_backward', action='store_true', default=False)
    parser.add_argument('--explicit_final', action='store_true', default=False)
    parser.add_argument('--marg', action='store_true', default=False, help='Whether marginalise out the Y path.')
    parser.add_argument('--id', type=int, default=666, help='The id of independent MC experiment.')
    parser.add_argument('--nchains', type=int, default=4, help='The number of MCMC chains.')
    parser.add_argument('--outdir', type=str, default='./toy/results')
    parser.add_argument('--quiet', action='store_true')
    args = parser.parse_args(argv)
    if args.marg:
        raise NotImplementedError('--marg routes through the closure tier (fbs_amd.samplers.gibbs_kernel with '
                                  'marg_y=True); this driver covers the fused configurations of the shipped scripts')
    dev = torch.device('cuda:0')
    key = ops.PRNGKey(args.id)                                                   # gp_gibbs.py:30

    # GP setting, gp_gibbs.py:33-58
    ell, sigma, d, obs_var = 1., 1., args.d, 1.
    zs = np.linspace(0., 5., d)
    cov_mat = sigma ** 2 * np.exp(-np.abs(zs[None, :] - zs[:, None]) / ell)
    key, subkey = ops.split(key)
    fs = np.linalg.cholesky(cov_mat) @ ops.normal(subkey, (d,), device=dev).cpu().numpy().astype(np.float64)
    key, subkey = ops.split(key)
    y0 = (fs + np.sqrt(obs_var) * ops.normal(subkey, (d,), device=dev).cpu().numpy()).astype(np.float32)
    Kyy = cov_mat + obs_var * np.eye(d)
    gp_mean = cov_mat @ np.linalg.solve(Kyy, y0.astype(np.float64))
    gp_cov = cov_mat - cov_mat @ np.linalg.solve(Kyy, cov_mat)
    joint_mean = np.zeros(2 * d)
    joint_cov = np.block([[cov_mat, cov_mat], [cov_mat, Kyy]])

    # SDE noising process, gp_gibbs.py:60-70
    T, nsteps = 1., 200
    ts = np.linspace(0, T, nsteps + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=4., t0=0., T=T) if args.sde == 'lin' \
        else StationaryConstLinearSDE(a=-0.5, b=1.)
    bridge = fbs_amd.LinearGaussianBridge(joint_mean, joint_cov, sde, ts, du=d, device=dev)
    nparticles, nsamples, nchains = args.nparticles, args.nsamples, args.nchains
    y0_t = torch.from_numpy(y0).to(dev)

    # Gibbs initial, gp_gibbs.py:153-162 (one bootstrap filter + backward smoother per chain)
    def gibbs_init(key_):
        key_fwd, key_bwd, key_bf = ops.split(key_, 3)
        vs = torch.flip(bridge.fwd_ys_sampler(key_fwd, y0_t), [0])
        uss = bootstrap_filter(bridge.transition_sampler, bridge.likelihood_logpdf, vs, ts, bridge.ref_sampler, key_bf,
                               nparticles, stratified, log=True, return_last=False)[0]
        return uss[-1, 0], bootstrap_backward_smoother(key_bwd, uss, vs, ts, bridge.transition_logpdf)

    key, subkey = ops.split(key)
    x0s = torch.stack([gibbs_init(k)[0] for k in ops.split(subkey, nchains)], 0)      # gp_gibbs.py:176-178
    bs_stars = np.zeros((nchains, nsteps + 1), np.int32)

    # Gibbs loop, gp_gibbs.py:180-190: per iteration key, subkey = split(key); key_chains = split(subkey, nchains)
    sweep = bridge.sweep_handle(nparticles, args.explicit_backward, args.explicit_final, nchains=nchains)
    key, x0s, bs_stars, samples = sweep.chain(key, x0s, y0, bs_stars, nsamples)
    gibbs_samples = samples.permute(1, 0, 2).cpu().numpy()                          # (nchains, nsamples, d)
    if not args.quiet:
        burn = min(100, nsamples // 2)
        err = np.abs(gibbs_samples[:, burn:].mean(axis=(0, 1)) - gp_mean).max()
        print(f'ID: {args.id} | Gibbs | {nchains} chains x {nsamples} sweeps | max |mean - gp_mean| = {err:.3f}')

    os.makedirs(args.outdir, exist_ok=True)
    out = os.path.join(args.outdir, f'gibbs{"-eb" if args.explicit_backward else ""}{"-ef" if args.explicit_final else ""}'
                       f'{"-marg" if args.marg else ""}-{args.sde}-{args.nparticles}-{args.id}')
    np.savez(out, samples=gibbs_samples, gp_mean=gp_mean, gp_cov=gp_cov)               # gp_gibbs.py:193-195
    return gibbs_samples, gp_mean, gp_cov


if __name__ == '__main__':
    main()
