"""Gaussian-process regression with the bootstrap-filter conditional sampler on MI355X.

Counterpart of the reference driver experiments/toy/gp_filter.py (same flags, key schedule and .npz schema:
samples (nsamples, d), gp_mean, gp_cov).  Every sample is one bootstrap_filter run (T = 200 steps); with the
analytic score the whole run is one hipGraph replay (for d > 16: drift on the f32 matrix cores)."""
import argparse
import os

import numpy as np
import torch

from _gp_toy import add_common_args, gp_setting
from fbs_amd import ops
from fbs_amd.samplers import bootstrap_filter, stratified


def main(argv=None):
    args = add_common_args(argparse.ArgumentParser()).parse_args(argv)
    dev = torch.device('cuda:0')
    g = gp_setting(args, dev)
    key, br, ts, y0 = g['key'], g['bridge'], g['ts'], g['y0_t']

    def conditional_sampler(key_):                                                  # gp_filter.py:134-142
        key_fwd, key_bwd, key_bf = ops.split(key_, 3)
        vs = torch.flip(br.fwd_ys_sampler(key_fwd, y0), [0])
        return bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, vs, ts, br.ref_sampler, key_bf,
                                args.nparticles, stratified, log=True, return_last=True)[0][0]

    samples = torch.empty((args.nsamples, g['d']), device=dev)
    for i in range(args.nsamples):                                                  # gp_filter.py:145-149
        key, subkey = ops.split(key)
        samples[i] = conditional_sampler(subkey)
    samples = samples.cpu().numpy()
    if not args.quiet:
        err = np.abs(samples.mean(axis=0) - g['gp_mean']).max()
        print(f'ID: {args.id} | filter | {args.nsamples} samples | max |mean - gp_mean| = {err:.3f}')
    os.makedirs(args.outdir, exist_ok=True)
    np.savez(os.path.join(args.outdir, f'filter-{args.sde}-{args.nparticles}-{args.id}'),
             samples=samples, gp_mean=g['gp_mean'], gp_cov=g['gp_cov'])             # gp_filter.py:152-153
    return samples, g['gp_mean'], g['gp_cov']


if __name__ == '__main__':
    main()
