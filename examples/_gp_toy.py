"""Shared setting of the Gaussian-process regression toy (experiments/toy/gp_*.py:30-70 of the reference):
GP prior on linspace(0, 5, d) with Matern-1/2 covariance, unit observation noise, the noising SDE and the
analytic-score bridge every driver of that family conditions with."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd  # noqa: E402
from fbs_amd import ops  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE  # noqa: E402


def add_common_args(parser):
    parser.add_argument('--d', type=int, default=10, help='The problem dimension.')
    parser.add_argument('--nparticles', type=int, default=10, help='The number of particles.')
    parser.add_argument('--nsamples', type=int, default=1000, help='The number of samples to draw.')
    parser.add_argument('--sde', type=str, default='const', help='The type of forward SDE.')
    parser.add_argument('--id', type=int, default=666, help='The id of independent MC experiment.')
    parser.add_argument('--outdir', type=str, default='./toy/results')
    parser.add_argument('--quiet', action='store_true')
    return parser


def gp_setting(args, dev):
    """-> dict(key, d, y0 (np f32), gp_mean, gp_cov, ts, sde, bridge).  Key schedule as gp_*.py:30-47."""
    key = ops.PRNGKey(args.id)
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
    joint_cov = np.block([[cov_mat, cov_mat], [cov_mat, Kyy]])
    T, nsteps = 1., 200
    ts = np.linspace(0, T, nsteps + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=4., t0=0., T=T) if args.sde == 'lin' \
        else StationaryConstLinearSDE(a=-0.5, b=1.)
    bridge = fbs_amd.LinearGaussianBridge(np.zeros(2 * d), joint_cov, sde, ts, du=d, device=dev)
    return dict(key=key, d=d, cov_mat=cov_mat, obs_var=obs_var, y0=y0, y0_t=torch.from_numpy(y0).to(dev), gp_mean=gp_mean, gp_cov=gp_cov, ts=ts, sde=sde,
                bridge=bridge, nsteps=nsteps)
