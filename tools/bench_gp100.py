"""The reference's own toy experiment (experiments/bashes/toy_gibbs.sh: gp_gibbs.py --d=100
--explicit_backward, 4 chains, T = 200): milliseconds per Gibbs sweep of all chains on the fused engine,
and (for comparison) on the closure tier the same call took before the matrix-core drift kernel existed."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fbs_amd  # noqa: E402
from fbs_amd import ops  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE  # noqa: E402
from helpers import toy_gp  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    d, T, C = 100, 200, 4
    toy = toy_gp(d)
    ts = np.linspace(0, 1.0, T + 1)
    br = fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(-0.5, 1.0), ts, d, device=dev)
    sizes = [int(a) for a in sys.argv[1:]] or [10, 100, 1000, 10000]
    for N in sizes:
        sw = br.sweep_handle(N, True, False, nchains=C)
        key = ops.PRNGKey(1)
        x0 = np.zeros((C, d), np.float32)
        bs = np.zeros((C, T + 1), np.int32)
        key, x0, bs, _ = sw.chain(key, x0, toy["y0"], bs, 3, keep=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get("GP100_SWEEPS", "20"))
        key, x0, bs, _ = sw.chain(key, x0, toy["y0"], bs, n, keep=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        flops = 2.0 * N * (2 * d) ** 2 * T * C
        print(f"d=100 N={N:6d} T={T} chains={C}: {dt * 1e3:8.3f} ms/sweep = {dt / T * 1e6:7.2f} us/step, "
              f"{N * T * C / dt:.3e} particle-steps/s, drift {flops / dt / 1e12:.2f} TFLOP/s (f32 MFMA peak 157)")
    # the particle filters of the same experiment family (gp_filter.py: one bootstrap filter per sample;
    # gp_pmcmc.py: one pmcmc_filter_step per MCMC iteration), nparticles = 100
    from fbs_amd.samplers import smc
    from fbs_amd.samplers import resampling as R
    N = 100
    y0 = torch.from_numpy(toy["y0"]).to(dev)
    vs = torch.flip(br.fwd_ys_sampler(ops.PRNGKey(5), y0), [0])
    init = ops.normal(ops.PRNGKey(6), (N, d), device=dev)
    for name, fn in (("bootstrap_filter", lambda k: smc.bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, vs, ts,
                                                                            lambda k_, v0, n_: init, k, N, R.stratified)),
                     ("pmcmc_filter_step", lambda k: smc.pmcmc_filter_step(k, vs, init, ts, br.transition_sampler,
                                                                            br.likelihood_logpdf, R.stratified, N))):
        fn(ops.PRNGKey(7))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            fn(ops.PRNGKey(8 + i))
        torch.cuda.synchronize()
        print(f"d=100 N={N} T={T} {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per run (one chain)")
    if len(sys.argv) > 1:
        return
    # closure tier for one chain at N = 100 (host loop, generic kernels)
    from fbs_amd.samplers import gibbs_kernel
    N = 100
    x0 = torch.zeros(d, device=dev)
    y0 = torch.from_numpy(toy["y0"]).to(dev)
    bs = np.zeros(T + 1, np.int32)

    def closure_sweep(k):
        return gibbs_kernel(k, x0, y0, None, bs, ts, br.fwd_sampler, br.sde, br.unpack, N, br.transition_sampler,
                            br.transition_logpdf, br.likelihood_logpdf, explicit_final=False, dummy_kw=None)
    try:
        closure_sweep(ops.PRNGKey(2))
    except TypeError:
        print("closure-tier comparison skipped (closures take no kwargs)")
        return
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        closure_sweep(ops.PRNGKey(3 + i))
    torch.cuda.synchronize()
    print(f"closure tier, one chain, N={N}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/sweep")


if __name__ == "__main__":
    main()
