"""Gaussian-process regression with particle-marginal Metropolis-Hastings on MI355X.

Counterpart of the reference driver experiments/toy/gp_pmcmc.py (same flags, key schedule and .npz schema:
samples (nchains, nsamples, d), gp_mean, gp_cov).  Every MCMC iteration of a chain is one
pmcmc_filter_step; with the analytic score it is one hipGraph replay (for d > 16: drift on the f32 matrix
cores).  The reference vmaps the chains; here they are iterated."""
import argparse
import os

import numpy as np
import torch

from _gp_toy import add_common_args, gp_setting
from fbs_amd import ops
from fbs_amd.samplers import bootstrap_filter, stratified
from fbs_amd.samplers.smc import pmcmc_kernel


def main(argv=None):
    parser = add_common_args(argparse.ArgumentParser())
    parser.add_argument('--delta', type=float, default=None, help='The pCN step size (None: independent proposals).')
    parser.add_argument('--nchains', type=int, default=4, help='The number of MCMC chains.')
    args = parser.parse_args(argv)
    dev = torch.device('cuda:0')
    g = gp_setting(args, dev)
    key, br, ts, y0 = g['key'], g['bridge'], g['ts'], g['y0_t']
    nchains, nsamples = args.nchains, args.nsamples

    def pmcmc_init(key_):                                                           # gp_pmcmc.py:145-151
        key_fwd, key_bwd, key_bf, key_ys = ops.split(key_, 4)
        vs = torch.flip(br.fwd_ys_sampler(key_fwd, y0), [0])
        x0s, log_ell = bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, vs, ts, br.ref_sampler, key_bf,
                                        args.nparticles, stratified, log=True, return_last=True)
        return x0s[0], log_ell, br.fwd_ys_sampler(key_ys, y0)

    key, subkey = ops.split(key)                                                    # gp_pmcmc.py:163-165
    state = [pmcmc_init(k) for k in ops.split(subkey, nchains)]
    samples = torch.empty((nchains, nsamples, g['d']), device=dev)
    accs = np.zeros(nsamples)
    for i in range(nsamples):                                                       # gp_pmcmc.py:170-179
        key, subkey = ops.split(key)
        for c, kc in enumerate(ops.split(subkey, nchains)):
            x0, log_ell, ys, mcmc_state = pmcmc_kernel(kc, *state[c], y0, ts, br.fwd_ys_sampler, g['sde'], br.ref_sampler,
                                                       br.transition_sampler, br.likelihood_logpdf, stratified,
                                                       args.nparticles, delta=args.delta)
            state[c] = (x0, log_ell, ys)
            samples[c, i] = x0
            if c == 0:
                accs[i] = float(mcmc_state.acceptance_prob)
    samples = samples.cpu().numpy()
    if not args.quiet:
        burn = min(100, nsamples // 2)
        err = np.abs(samples[:, burn:].mean(axis=(0, 1)) - g['gp_mean']).max()
        print(f'ID: {args.id} | pMCMC | {nchains} chains x {nsamples} iterations | mean acceptance {accs.mean():.3f} '
              f'| max |mean - gp_mean| = {err:.3f}')
    os.makedirs(args.outdir, exist_ok=True)
    np.savez(os.path.join(args.outdir, f'pmcmc-{args.delta}-{args.sde}-{args.nparticles}-{args.id}'),
             samples=samples, gp_mean=g['gp_mean'], gp_cov=g['gp_cov'])             # gp_pmcmc.py:186-187
    return samples, g['gp_mean'], g['gp_cov']


if __name__ == '__main__':
    main()
