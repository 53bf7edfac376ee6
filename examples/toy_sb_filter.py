"""Gaussian-process regression with the bootstrap-filter conditional sampler on the NON-separable Gaussian Schrodinger bridge.

Counterpart of the reference driver experiments/sb/filter.py (same flags incl. `--x0 proper|heuristic`, key schedule and .npz
schema `filter-<x0>-<nparticles>-<id>.npz`): the forward observation path is an Euler-Maruyama simulation of the bridge's
drift from (x0, y0) with x0 a GP-posterior draw ('proper') or N(0, I) ('heuristic'); every sample is one bootstrap_filter run
on the closure tier (resampling, gathers, normalisation: libfbsmi kernels)."""
import argparse
import os

import numpy as np
import torch

from toy_sb_gibbs import common_args, sb_setting
from fbs_amd import ops
from fbs_amd.samplers import bootstrap_filter, stratified
from fbs_amd.sdes import euler_maruyama


def main(argv=None):
    parser = common_args(argparse.ArgumentParser())
    parser.add_argument('--x0', type=str, default='heuristic', help="How the forward path's x0 is drawn: 'proper' or 'heuristic'.")
    args = parser.parse_args(argv)
    if args.x0 not in ('proper', 'heuristic'):
        raise ValueError(f'Invalid "{args.x0}" method')
    dev = torch.device('cuda:0')
    g = sb_setting(args, dev)
    key, d = g.key, g.d

    def fwd_ys_sampler(key_):                                                        # sb/filter.py:137-148
        key_x0, key_em = ops.split(key_)
        x0_ = g.gp_posterior_sampler(key_x0) if args.x0 == 'proper' else ops.normal(key_x0, (d,), device=dev)
        xy0 = torch.cat([x0_, g.y0])
        return euler_maruyama(key_em, xy0, g.ts, g.drift, lambda t: 1., integration_nsteps=10, return_path=True)[:, d:]

    def conditional_sampler(key_):                                                   # :152-164
        key_fwd, key_bwd, key_bf = ops.split(key_, 3)
        vs = torch.flip(fwd_ys_sampler(key_fwd), [0])
        return bootstrap_filter(g.transition_sampler, g.likelihood_logpdf, vs, g.ts, g.ref_sampler, key_bf, args.nparticles,
                                stratified, log=True, return_last=True)[0][0]

    samples = torch.empty((args.nsamples, d), device=dev)
    for i in range(args.nsamples):                                                   # :167-172
        key, subkey = ops.split(key)
        samples[i] = conditional_sampler(subkey)
    samples = samples.cpu().numpy()
    if not args.quiet:
        err = np.abs(samples.mean(axis=0) - g.gp_mean).max()
        print(f'ID: {args.id} | SB filter ({args.x0}) | {args.nsamples} samples | max |mean - gp_mean| = {err:.3f}')
    os.makedirs(args.outdir, exist_ok=True)
    np.savez(os.path.join(args.outdir, f'filter-{args.x0}-{args.nparticles}-{args.id}'),
             samples=samples, gp_mean=g.gp_mean, gp_cov=g.gp_cov)                    # :175-176
    return samples, g.gp_mean, g.gp_cov


if __name__ == '__main__':
    main()
