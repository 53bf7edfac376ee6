"""GPU parity of the linear-Gaussian path: model closures, the fused Gibbs sweep and the
closure-driven (generic) tier, each against the CPU oracle on the same keys.

Bar (BASELINE.json north_star): ancestor indices bit-exact; float32 states / weights within 1e-5
relative -- in fact these tests demand bit equality, since both sides evaluate
include/fbsmi_math.h and the same summation tree.
"""
import numpy as np
import pytest
import torch

from helpers import toy_2d, toy_4d, toy_31, toy_gp, oracle_model_from

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _bridge(toy, ts, dev, sde=None):
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    sde = sde or StationaryConstLinearSDE(a=-0.5, b=1.)
    return fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], sde, ts, toy["du"], device=dev)


def _eq(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    if a.dtype == np.float32:
        bad = np.flatnonzero(a.view(np.uint32).ravel() != b.view(np.uint32).ravel())
    else:
        bad = np.flatnonzero(a.ravel() != b.ravel())
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.ravel()[bad[:5]]} vs {b.ravel()[bad[:5]]}"


@pytest.mark.parametrize("toy", [toy_2d, toy_4d, toy_31])
def test_closures_bit_exact(toy, oracle, dev):
    toy = toy()
    ts = np.linspace(0, 1, 51)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(0)
    n = 1000
    us_prev = rng.normal(size=(n, br.du)).astype(np.float32)
    v_prev = rng.normal(size=br.dv).astype(np.float32)
    v = rng.normal(size=br.dv).astype(np.float32)
    u = rng.normal(size=br.du).astype(np.float32)
    key = oracle.PRNGKey(3)
    for k in (0, 17, 49):
        t_prev = ts[k]
        up_t = torch.from_numpy(us_prev).to(dev)
        _eq(_np(br.transition_sampler(up_t, torch.from_numpy(v_prev).to(dev), t_prev, key)),
            oracle.lg_transition_sampler(om, k, us_prev, v_prev, key), "transition_sampler")
        _eq(_np(br.likelihood_logpdf(torch.from_numpy(v).to(dev), up_t, torch.from_numpy(v_prev).to(dev), t_prev)),
            oracle.lg_likelihood_logpdf(om, k, v, us_prev, v_prev), "likelihood_logpdf")
        _eq(_np(br.transition_logpdf(torch.from_numpy(u).to(dev), up_t, torch.from_numpy(v_prev).to(dev), t_prev)),
            oracle.lg_transition_logpdf(om, k, u, us_prev, v_prev), "transition_logpdf")
    xy0 = np.concatenate([rng.normal(size=br.du), toy["y0"]]).astype(np.float32)
    _eq(_np(br.fwd_sampler(key, torch.from_numpy(xy0[:br.du]).to(dev), torch.from_numpy(xy0[br.du:]).to(dev))),
        oracle.lg_fwd_sampler(om, key, xy0), "fwd_sampler")


CASES = [
    # toy, N, T, Tend, eb, ef
    (toy_2d, 10, 100, 1.0, True, False),      # the reference's own test_gibbs configuration
    (toy_2d, 10, 100, 1.0, False, False),
    (toy_2d, 10, 40, 4.0, True, True),
    (toy_2d, 10, 40, 4.0, False, True),
    (toy_2d, 1024, 200, 1.0, True, False),    # BASELINE config 1
    (toy_2d, 777, 30, 1.0, True, False),      # ragged: not a multiple of the tile
    (toy_2d, 1, 20, 1.0, True, False),        # single particle
    (toy_2d, 2, 20, 1.0, False, False),
    (toy_4d, 300, 50, 1.0, True, False),
    (toy_4d, 300, 50, 2.0, False, True),
    (toy_31, 513, 25, 1.0, True, False),
    (toy_2d, 200000, 12, 1.0, True, False),   # ITEMS = 4 kernels
    (toy_4d, 256, 9, 1.0, True, False),       # the largest single-launch-per-step ensemble
    (toy_4d, 255, 9, 1.0, False, True),       # ... with explicit_final: N + 1 = 256 slots
    (toy_31, 257, 9, 1.0, True, False),       # just past it: three launches per step, two tiles
    (toy_4d, 512, 12, 1.0, True, False),      # powers of two from two tiles up: the two-launch step (k_lg_prop1t)
    (toy_31, 2048, 8, 1.0, False, False),     # ... with the stored path
    (toy_2d, 4095, 8, 1.0, True, True),       # ... explicit_final: N + 1 = 4096 slots
    (toy_2d, 32768, 6, 1.0, True, False),
    (toy_4d, 512, 9, 1.0, True, True),        # explicit_final on a power of two: N + 1 = 2^k + 1 slots, still two launches
    (toy_2d, 4096, 8, 1.0, True, True),       # ... (the tree of the first 2^k slots + an extra one-slot tile)
    (toy_31, 1024, 6, 1.0, False, True),      # ... with the stored path
    (toy_2d, 65536, 4, 1.0, True, True),      # ... BASELINE config 2's ensemble with explicit_final: 65 537 slots
]


@pytest.mark.parametrize("toy,N,T,Tend,eb,ef", CASES)
def test_fused_sweep_matches_oracle(toy, N, T, Tend, eb, ef, oracle, dev):
    toy = toy()
    ts = np.linspace(0, Tend, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(N + T)
    x0 = rng.normal(size=br.du).astype(np.float32)
    bs = rng.integers(0, N, T + 1).astype(np.int32)
    sweep = br.sweep_handle(N, eb, ef)
    for trial, use_graph in enumerate((False, True, True)):
        key = oracle.split(oracle.PRNGKey(42 + trial), 2)[1]
        want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, eb, ef, debug=True)
        got = sweep.sweep(key, x0, toy["y0"], bs, use_graph=use_graph)
        v = sweep.views()
        _eq(_np(v["us_T"]), want[4], "final particles")
        _eq(_np(v["lw_T"]), want[5], "final log-weights")
        _eq(_np(got[0]), want[0], "x0_next")
        _eq(_np(got[1]), want[1], "us_star_next")
        _eq(_np(got[2]), want[2], "bs_star_next")
        _eq(_np(got[3]), want[3], "acc")
        # chain the state like the reference's driver does
        x0, bs = want[0], want[2]


WIDE_CASES = [
    # du, dv, N, T, eb : wide models run the drift on the matrix cores (k_lgw_prop)
    (100, 100, 10, 12, True),     # the reference's toy_gibbs.sh configuration (d = 100), few particles
    (100, 100, 100, 6, True),
    (20, 20, 300, 10, True),      # more than one logsumexp tile
    (33, 17, 77, 8, True),        # odd sizes: D = 50 is not a multiple of 4, row tiles straddle du
    (17, 5, 40, 8, False),        # stored path + backward scanning
    (128, 128, 33, 3, True),      # the largest supported model
    (24, 24, 1, 4, True),         # a single particle
    (24, 24, 2, 4, False),
    (40, 40, 32, 4, True),        # exactly one slot tile
    (40, 40, 256, 4, True),       # the largest single-launch-per-step ensemble
    (40, 40, 257, 4, True),       # just past it: five launches per step
    (30, 18, 700, 5, False),      # several logsumexp tiles, stored path
    (20, 20, 40000, 3, True),     # enough workgroups for k_lgw_gemm_fat (all row tiles per 32-slot workgroup)
    (33, 17, 35000, 2, False),    # ... with odd sizes and the stored path
    (100, 100, 12000, 2, True),   # ... at the reference's d = 100 (seven row tiles)
    (100, 100, 3300, 2, True),    # just past the switch to k_lgw_gemm_fat (728 tiled workgroups), a partly filled last slot tile
    (100, 100, 2500, 2, False),   # just below it (tiled kernel, stored path)
]


@pytest.mark.parametrize("du,dv,N,T,eb", WIDE_CASES)
def test_wide_fused_sweep_matches_oracle(du, dv, N, T, eb, oracle, dev):
    toy = toy_gp(du, dv)
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(du + N)
    x0 = rng.normal(size=du).astype(np.float32)
    bs = rng.integers(0, N, T + 1).astype(np.int32)
    sweep = br.sweep_handle(N, eb, False)
    for trial, use_graph in enumerate((False, True)):
        key = oracle.split(oracle.PRNGKey(7 + trial), 2)[1]
        want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, eb, False, debug=True)
        got = sweep.sweep(key, x0, toy["y0"], bs, use_graph=use_graph)
        v = sweep.views()
        _eq(_np(v["us_T"]), want[4], "final particles")
        _eq(_np(v["lw_T"]), want[5], "final log-weights")
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i]), want[i], what)
        x0, bs = want[0], want[2]


def test_wide_handles_of_different_widths_coexist(oracle, dev):
    """The dynamic-LDS limit of the drift kernels is a per-function attribute: creating a handle for a small model
    must not lower it under a live handle of a large one."""
    T, N = 3, 33
    ts = np.linspace(0, 1.0, T + 1)
    big, small = toy_gp(128), toy_gp(20)
    br_big, br_small = _bridge(big, ts, dev), _bridge(small, ts, dev)
    h_big = br_big.sweep_handle(N, True, False)
    h_small = br_small.sweep_handle(N, True, False)
    key = oracle.split(oracle.PRNGKey(11), 2)[0]
    for toy, br, h in ((small, br_small, h_small), (big, br_big, h_big)):
        rng = np.random.default_rng(toy["du"])
        x0 = rng.normal(size=toy["du"]).astype(np.float32)
        bs = rng.integers(0, N, T + 1).astype(np.int32)
        want = oracle.gibbs_kernel_lg(oracle_model_from(oracle, br), key, x0, toy["y0"], bs, N, True, False)
        got = h.sweep(key, x0, toy["y0"], bs, use_graph=False)
        _eq(_np(got[0]), want[0], "x0_next")
        _eq(_np(got[1]), want[1], "us_star_next")


def test_wide_batched_chains_match_single_chains(oracle, dev):
    toy = toy_gp(40)
    T, N, C = 6, 50, 3
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    rng = np.random.default_rng(1)
    x0 = rng.normal(size=(C, 40)).astype(np.float32)
    bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
    keys = oracle.split(oracle.PRNGKey(3), C)
    batch = br.sweep_handle(N, True, False, nchains=C).sweep(keys, x0, toy["y0"], bs)
    one = br.sweep_handle(N, True, False)
    for c in range(C):
        got = one.sweep(keys[c], x0[c], toy["y0"], bs[c])
        for i in range(4):
            _eq(_np(batch[i][c]), _np(got[i]), f"chain {c} output {i}")


@pytest.mark.parametrize("d,C,N,T", [(100, 1, 100, 7), (24, 1, 40, 9), (40, 2, 100, 6), (24, 4, 300, 5), (20, 1, 3000, 3)])
def test_wide_chained_sweeps_match_oracle(d, C, N, T, oracle, dev):
    """The driver loop (gp_gibbs.py:182-187) on wide models: sweeps chained inside the engine -- the one-launch step's noise
    is drawn one step AHEAD (by idle / extra blocks of the previous launch, across the sweep boundary by the sweep's first
    kernels), a single chain's launches are pinned to one XCD -- against the oracle's chain, sample for sample."""
    toy = toy_gp(d)
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(d + C + N)
    nsw = 4
    if C == 1:
        x0 = rng.normal(size=d).astype(np.float32)
        bs = rng.integers(0, N, T + 1).astype(np.int32)
        sweep = br.sweep_handle(N, True, False)
        key, x0f, bsf, x0s = sweep.chain(oracle.PRNGKey(31), x0, toy["y0"], bs, nsw)
        okey, ox0, obs, oout = oracle.gibbs_chain_lg(om, oracle.PRNGKey(31), x0, toy["y0"], bs, N, nsw)
    else:
        x0 = rng.normal(size=(C, d)).astype(np.float32)
        bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
        sweep = br.sweep_handle(N, True, False, nchains=C)
        key, x0f, bsf, x0s = sweep.chain(oracle.PRNGKey(31), x0, toy["y0"], bs, nsw)
        okey, ox0, obs, oout = oracle.gibbs_chains_lg(om, oracle.PRNGKey(31), x0, toy["y0"], bs, N, nsw)
    np.testing.assert_array_equal(key, okey)
    _eq(_np(x0s).reshape(np.asarray(oout).shape), oout, "chain samples")
    _eq(_np(bsf).reshape(np.asarray(obs).shape), obs, "final bs_star")


@pytest.mark.parametrize("toy,N,T", [(toy_2d, 64, 30), (toy_4d, 100, 20)])
def test_fused_forward_pass_paths_match_oracle(toy, N, T, oracle, dev):
    """As / uss / log_wss of csmc.forward_pass (csmc.py:161-164) on the stored-path variant."""
    toy = toy()
    ts = np.linspace(0, 1, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(1)
    x0 = rng.normal(size=br.du).astype(np.float32)
    bs = rng.integers(0, N, T + 1).astype(np.int32)
    key = oracle.PRNGKey(8)
    sweep = br.sweep_handle(N, False, False)
    sweep.sweep(key, x0, toy["y0"], bs, use_graph=False)
    v = sweep.views()
    # rebuild the oracle's forward pass inputs the way gibbs_kernel does (gibbs.py:126-144)
    k_fwd, k_csmc, _ = oracle.split(key, 3)
    path = oracle.lg_fwd_sampler(om, k_fwd, np.concatenate([x0, toy["y0"]]))
    us, vs = path[::-1, :br.du].copy(), path[::-1, br.du:].copy()
    _eq(_np(v["us_star"]), us, "us_star")
    _eq(_np(v["vs"]), vs, "vs")
    k_csmc_fwd = oracle.split(k_csmc, 2)[0]
    us0 = np.tile(us[0], (N, 1)).astype(np.float32)
    lw0 = np.full(N, np.float32(-np.log(N)), np.float32)
    fp = oracle.csmc_forward_pass_lg(om, k_csmc_fwd, us, bs, vs, us0, lw0)
    _eq(_np(v["As"]), fp["As"], "As")
    _eq(_np(v["uss"]), fp["uss"], "uss")
    _eq(_np(v["log_wss"]), fp["log_wss"], "log_wss")


@pytest.mark.parametrize("eb,ef", [(True, False), (False, False), (True, True)])
def test_generic_tier_matches_oracle(eb, ef, oracle, dev):
    """fbs_amd.samplers.gibbs_kernel driven through the Python closure protocol (host loop)."""
    from fbs_amd.samplers import gibbs_kernel
    from fbs_amd.samplers.csmc.csmc import forward_pass
    toy = toy_2d()
    T, N = 25, 50
    ts = np.linspace(0, 2, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(4)
    x0 = rng.normal(size=1).astype(np.float32)
    bs = rng.integers(0, N, T + 1).astype(np.int32)
    key = oracle.PRNGKey(77)
    # plain lambdas hide the descriptor, so the dispatcher must take the generic tier
    ts_t = ts
    got = gibbs_kernel(key, torch.from_numpy(x0).to(dev), torch.from_numpy(toy["y0"]).to(dev), None, bs, ts_t,
                       lambda k, a, b: br.fwd_sampler(k, a, b), br.sde, lambda xy: br.unpack(xy), N,
                       lambda *a: br.transition_sampler(*a), lambda *a: br.transition_logpdf(*a),
                       lambda *a: br.likelihood_logpdf(*a), marg_y=False, explicit_backward=eb, explicit_final=ef)
    want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, eb, ef)
    _eq(_np(got[0]), want[0], "x0")
    _eq(_np(got[1]), want[1], "us_star")
    _eq(_np(got[2]), want[2], "bs_star")
    _eq(_np(got[3]), want[3], "acc")
    # and the dispatcher's fused route gives the same answer
    got2 = gibbs_kernel(key, x0, toy["y0"], None, bs, ts_t, br.fwd_sampler, br.sde, br.unpack, N,
                        br.transition_sampler, br.transition_logpdf, br.likelihood_logpdf, marg_y=False,
                        explicit_backward=eb, explicit_final=ef)
    for a, b, w in zip(got2, want, ("x0", "us_star", "bs_star", "acc")):
        _eq(_np(a), b, "fused " + w)


def test_chain_and_closed_form_posterior(oracle, dev):
    """tests/test_gibbs.py:16-123 of the reference: N=10, 100 steps, 10 000 sweeps; the chain must
    target p(x0 | y0) = N(-1.8, 1.68) within the reference's own tolerances, and agree with the
    oracle's chain bit for bit."""
    toy = toy_2d()
    ts = np.linspace(0, 1, 101)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    sweep = br.sweep_handle(10, True, False)
    nsweeps = 10000
    key, x0, bs, x0s = sweep.chain(oracle.PRNGKey(666), np.zeros(1, np.float32), toy["y0"], np.zeros(101, np.int32),
                                   nsweeps)
    okey, ox0, obs, ox0s = oracle.gibbs_chain_lg(om, oracle.PRNGKey(666), [0.], toy["y0"], np.zeros(101, np.int32),
                                                 10, nsweeps)
    _eq(_np(x0s), ox0s, "chain samples")
    np.testing.assert_array_equal(key, okey)
    _eq(_np(bs), obs, "final bs_star")
    xs = _np(x0s)[10:, 0].astype(np.float64)
    np.testing.assert_allclose(xs.mean(), -1.8, rtol=5e-2)      # test_gibbs.py:122
    np.testing.assert_allclose(xs.var(), 1.68, rtol=2e-2)       # test_gibbs.py:123


@pytest.mark.parametrize("toy,C,N,T", [(toy_2d, 4, 100, 30), (toy_4d, 3, 300, 12), (toy_2d, 2, 1024, 20),
                                       (toy_2d, 5, 300, 12), (toy_4d, 7, 512, 8)])   # five and more chains: three groups of unequal sizes
def test_batched_chains_match_oracle(toy, C, N, T, oracle, dev):
    """nchains > 1: the reference's jax.vmap over chains (gp_gibbs.py:172-187), one launch sequence (per chain group)."""
    toy = toy()
    ts = np.linspace(0, 1, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(C * N)
    x0 = rng.normal(size=(C, br.du)).astype(np.float32)
    bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
    sweep = br.sweep_handle(N, True, False, nchains=C)
    # single batched sweep with explicit per-chain keys
    keys = oracle.split(oracle.PRNGKey(5), C)
    got = sweep.sweep(keys, x0, toy["y0"], bs)
    v = sweep.views()
    for c in range(C):
        want = oracle.gibbs_kernel_lg(om, keys[c], x0[c], toy["y0"], bs[c], N, True, False, debug=True)
        _eq(_np(got[0][c]), want[0], f"x0 chain {c}")
        _eq(_np(got[1][c]), want[1], f"us_star chain {c}")
        _eq(_np(got[2][c]), want[2], f"bs chain {c}")
        _eq(_np(got[3][c]), want[3], f"acc chain {c}")
        _eq(_np(v["us_T"][c]), want[4], f"particles chain {c}")
        _eq(_np(v["lw_T"][c]), want[5], f"log-weights chain {c}")
    # chained sweeps with the gp_gibbs key schedule
    nsw = 6
    key, x0f, bsf, x0s = sweep.chain(oracle.PRNGKey(9), x0, toy["y0"], bs, nsw)
    okey, ox0, obs, oout = oracle.gibbs_chains_lg(om, oracle.PRNGKey(9), x0, toy["y0"], bs, N, nsw)
    np.testing.assert_array_equal(key, okey)
    _eq(_np(x0s), oout, "chain samples")
    _eq(_np(bsf), obs, "final bs_star")


def test_baseline_config2_one_sweep(oracle, dev):
    """BASELINE config 2: N = 65 536, T = 500, ts = linspace(0, 2, 501)."""
    toy = toy_2d()
    N, T = 65536, 500
    ts = np.linspace(0, 2, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    bs = np.zeros(T + 1, np.int32)
    key = oracle.split(oracle.PRNGKey(666), 2)[1]
    sweep = br.sweep_handle(N, True, False)
    got = sweep.sweep(key, np.zeros(1, np.float32), toy["y0"], bs)
    v = sweep.views()
    want = oracle.gibbs_kernel_lg(om, key, np.zeros(1, np.float32), toy["y0"], bs, N, True, False, debug=True)
    _eq(_np(v["us_T"]), want[4], "final particles")
    _eq(_np(v["lw_T"]), want[5], "final log-weights")
    _eq(_np(got[0]), want[0], "x0_next")
    _eq(_np(got[2]), want[2], "bs_star_next")
    # size-independent properties: weights normalised, indices in range
    w = np.exp(_np(v["lw_T"]).astype(np.float64))
    assert abs(w.sum() - 1) < 1e-4
    b = _np(got[2])
    assert b.min() >= 0 and b.max() < N


def test_toy_driver_targets_the_gp_posterior(tmp_path, dev):
    """examples/toy_gibbs.py (counterpart of experiments/toy/gp_gibbs.py, d = 10, joint dimension 20,
    4 vmapped chains, explicit backward): the chains must recover the GP-regression posterior, the
    closed form the reference's tabulators compare against (tabulate_toy.py:46-52)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "toy_gibbs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "toy_gibbs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    samples, gp_mean, gp_cov = mod.main(["--d", "10", "--nparticles", "100", "--nsamples", "1500", "--explicit_backward",
                                         "--nchains", "4", "--outdir", str(tmp_path), "--quiet"])
    assert samples.shape == (4, 1500, 10)
    x = samples[:, 200:].reshape(-1, 10).astype(np.float64)
    assert np.abs(x.mean(0) - gp_mean).max() < 0.12
    assert np.abs(np.diag(np.cov(x.T)) - np.diag(gp_cov)).max() < 0.1
    saved = np.load(os.path.join(str(tmp_path), "gibbs-eb-const-100-666.npz"))
    assert set(saved.files) == {"samples", "gp_mean", "gp_cov"}


def _load_example(name):
    import importlib.util
    import os
    import sys
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")
    if ex not in sys.path:
        sys.path.insert(0, ex)
    spec = importlib.util.spec_from_file_location(name, os.path.join(ex, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_toy_filter_and_pmcmc_drivers(tmp_path, dev):
    """examples/toy_filter.py / toy_pmcmc.py (counterparts of experiments/toy/gp_filter.py / gp_pmcmc.py) at
    d = 20 (a wide model: fused filters with the drift on the matrix cores): the bootstrap-filter sampler is
    biased but close for 200 particles, pMCMC targets the GP posterior exactly."""
    import os
    filt = _load_example("toy_filter")
    samples, gp_mean, gp_cov = filt.main(["--d", "20", "--nparticles", "200", "--nsamples", "300", "--outdir", str(tmp_path),
                                          "--quiet"])
    assert samples.shape == (300, 20) and np.isfinite(samples).all()
    assert np.abs(samples.mean(0) - gp_mean).max() < 0.35
    assert set(np.load(os.path.join(str(tmp_path), "filter-const-200-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}
    pm = _load_example("toy_pmcmc")
    samples, gp_mean, gp_cov = pm.main(["--d", "20", "--nparticles", "200", "--nsamples", "150", "--nchains", "2",
                                        "--delta", "0.005", "--outdir", str(tmp_path), "--quiet"])
    assert samples.shape == (2, 150, 20) and np.isfinite(samples).all()
    # pCN with a small step mixes slowly: the chains stay in the bulk of the posterior (a loose check), the
    # acceptance machinery and the .npz schema are what this test pins
    z = (samples[:, 50:].reshape(-1, 20).mean(0) - gp_mean) / np.sqrt(np.diag(gp_cov))
    assert np.abs(z).max() < 2.5
    assert set(np.load(os.path.join(str(tmp_path), "pmcmc-0.005-const-200-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}


@pytest.mark.parametrize("tree", ["1", "0", "one-tile-workgroups", "four-tile-workgroups"])
def test_tree_step_reference_indices_on_tile_edges(tree, oracle, dev, monkeypatch):
    """The two-launch step (searches walk the summation tree) and the three-launch step it replaces (FBSMI_TREE_STEP=0)
    with the reference path sitting on tile boundaries: first / last element of a tile, of the ensemble."""
    monkeypatch.setenv("FBSMI_TREE_STEP", "0" if tree == "0" else "1")
    if tree == "one-tile-workgroups":   # k_lg_prop1t<., 1> instead of the 512-thread workgroups that own two tiles
        monkeypatch.setenv("FBSMI_TREE_HALVES", "1")
    if tree == "four-tile-workgroups":
        monkeypatch.setenv("FBSMI_TREE_HALVES", "4")
    toy = toy_2d()
    N, T = 1024, 12
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    x0 = np.array([0.3], np.float32)
    bs = np.array([0, 256, 255, 1023, 512, 511, 0, 768, 1, 1023, 256, 257, 0], np.int32)
    sweep = br.sweep_handle(N, True, False)
    for trial in range(2):
        key = oracle.split(oracle.PRNGKey(5 + trial), 2)[0]
        want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, True, False, debug=True)
        got = sweep.sweep(key, x0, toy["y0"], bs, use_graph=bool(trial))
        v = sweep.views()
        _eq(_np(v["us_T"]), want[4], "final particles")
        _eq(_np(v["lw_T"]), want[5], "final log-weights")
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i]), want[i], what)


@pytest.mark.parametrize("plus1", ["1", "0"])
def test_power_of_two_plus_one_slots_with_the_reference_on_the_last_slot(plus1, oracle, dev, monkeypatch):
    """N = 2^k + 1 rows (explicit_final): the two-launch step walks the tree of the first 2^k slots, the last slot is an extra
    tile.  Reference indices on the last slot itself (its J_prob leaf is the one that changes), on tile edges, and several
    chains; FBSMI_TREE_PLUS1=0 keeps the three-launch step for comparison."""
    monkeypatch.setenv("FBSMI_TREE_PLUS1", plus1)
    toy = toy_2d()
    n, T, C = 2048, 10, 3
    N = n + 1
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(17)
    x0 = rng.normal(size=(C, br.du)).astype(np.float32)
    bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
    bs[0] = [N - 1, 0, N - 1, N - 1, 255, 256, N - 2, N - 1, 1024, 0, N - 1]
    keys = oracle.split(oracle.PRNGKey(29), C)
    sweep = br.sweep_handle(n, True, True, nchains=C)
    got = sweep.sweep(keys, x0, toy["y0"], bs)
    v = sweep.views()
    for c in range(C):
        want = oracle.gibbs_kernel_lg(om, keys[c], x0[c], toy["y0"], bs[c], n, True, True, debug=True)
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i][c]), want[i], f"{what} chain {c}")
        _eq(_np(v["us_T"][c]), want[4], f"particles chain {c}")
        _eq(_np(v["lw_T"][c]), want[5], f"log-weights chain {c}")
    br._sweeps.clear()


@pytest.mark.parametrize("tiles,toy,N,T,C", [("2", toy_4d, 4096, 6, 2), ("4", toy_31, 8192, 5, 3), ("", toy_2d, 65536, 3, 4)])
def test_tree_step_workgroups_owning_several_tiles(tiles, toy, N, T, C, oracle, dev, monkeypatch):
    """k_lg_prop1t<., 2 | 4>: 512 / 1024-thread workgroups own adjacent tiles and only the first 256 threads build the trees
    and find J.  Forced on mid-sized ensembles (several workgroups per chain), and by its own rule at BASELINE config 2's
    size with the reference's four chains."""
    if tiles:
        monkeypatch.setenv("FBSMI_TREE_HALVES", tiles)
    toy_ = toy()
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy_, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(N + C)
    x0 = rng.normal(size=(C, br.du)).astype(np.float32)
    bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
    keys = oracle.split(oracle.PRNGKey(23), C)
    sweep = br.sweep_handle(N, True, False, nchains=C)
    got = sweep.sweep(keys, x0, toy_["y0"], bs)
    v = sweep.views()
    for c in range(C):
        want = oracle.gibbs_kernel_lg(om, keys[c], x0[c], toy_["y0"], bs[c], N, True, False, debug=True)
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i][c]), want[i], f"{what} chain {c}")
        _eq(_np(v["us_T"][c]), want[4], f"particles chain {c}")
        _eq(_np(v["lw_T"][c]), want[5], f"log-weights chain {c}")


@pytest.mark.parametrize("tree", ["1", "0"])
def test_two_slot_prop_kernel_matches_oracle(tree, oracle, dev, monkeypatch):
    """k_lg_prop2 / k_lg_prop2t (two slots per thread, N/2 apart: the kill-test / redraw / noise draws of both slots
    from three Threefry calls) are chosen for big batches; forced here on small ones, with the two-launch step (powers
    of two) and without.  Bit-exact like every other variant."""
    monkeypatch.setenv("FBSMI_TWO_SLOT_PROP", "1")
    monkeypatch.setenv("FBSMI_TREE_STEP", tree)
    for toy, N, T, C in ((toy_2d, 1024, 40, 1), (toy_4d, 512, 12, 3), (toy_31, 2048, 6, 2), (toy_2d, 1536, 8, 2),
                         (toy_2d, 65536, 6, 2)):
        toy_ = toy()
        ts = np.linspace(0, 1.0, T + 1)
        br = _bridge(toy_, ts, dev)
        om = oracle_model_from(oracle, br)
        rng = np.random.default_rng(N + T)
        x0 = rng.normal(size=(C, br.du)).astype(np.float32)
        bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
        keys = oracle.split(oracle.PRNGKey(11), C)
        sweep = br.sweep_handle(N, True, False, nchains=C)
        got = sweep.sweep(keys if C > 1 else keys[0], x0 if C > 1 else x0[0], toy_["y0"], bs if C > 1 else bs[0])
        for c in range(C):
            want = oracle.gibbs_kernel_lg(om, keys[c], x0[c], toy_["y0"], bs[c], N, True, False)
            for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
                _eq(_np(got[i][c] if C > 1 else got[i]), want[i], f"N={N} chain {c} {what}")


@pytest.mark.parametrize("variant", ["queue", "generic"])
@pytest.mark.parametrize("toy,N,T,C,eb", [(toy_2d, 200000, 5, 2, True), (toy_4d, 300001, 4, 1, False), (toy_2d, 1100000, 3, 1, True),
                                         (toy_31, 1048576 + 4096, 2, 1, True), (toy_2d, 1100003, 2, 2, True)])
def test_several_slots_per_thread_kernels(variant, toy, N, T, C, eb, oracle, dev, monkeypatch):
    """N > 131072: tiles of 1024 / 4096 slots (ITEMS = 4 / 16).  k_lg_heaps + k_lg_propQ (lane-major slots, kill tests first,
    the killed sources' searches compacted through an LDS queue, compact heaps) and the one-slot-after-the-other kernel it
    replaced (FBSMI_GENERIC_PROP=1): ragged last tiles, the stored path, several chains, du = 3.  Bit-exact."""
    monkeypatch.setenv("FBSMI_GENERIC_PROP", "1" if variant == "generic" else "0")
    toy_ = toy()
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy_, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(N + T)
    x0 = rng.normal(size=(C, br.du)).astype(np.float32)
    bs = rng.integers(0, N, (C, T + 1)).astype(np.int32)
    bs[0, 1] = N - 1
    bs[0, 2] = 0
    keys = oracle.split(oracle.PRNGKey(31), C)
    sweep = br.sweep_handle(N, eb, False, nchains=C)
    got = sweep.sweep(keys if C > 1 else keys[0], x0 if C > 1 else x0[0], toy_["y0"], bs if C > 1 else bs[0])
    v = sweep.views()
    for c in range(C):
        want = oracle.gibbs_kernel_lg(om, keys[c], x0[c], toy_["y0"], bs[c], N, eb, False, debug=True)
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i][c] if C > 1 else got[i]), want[i], f"N={N} chain {c} {what}")
        _eq(_np(v["us_T"][c] if C > 1 else v["us_T"]), want[4], f"particles chain {c}")
        _eq(_np(v["lw_T"][c] if C > 1 else v["lw_T"]), want[5], f"log-weights chain {c}")
    del sweep
    br._sweeps.clear()


def test_toy_sb_gibbs_driver(tmp_path, dev):
    """examples/toy_sb_gibbs.py (counterpart of experiments/sb/gibbs.py: Gibbs on the non-separable Gaussian
    Schrodinger bridge, closure tier with Euler-Maruyama forward paths): runs, stays in the bulk of the GP posterior,
    writes the reference's .npz schema."""
    import os
    mod = _load_example("toy_sb_gibbs")
    samples, gp_mean, gp_cov = mod.main(["--d", "3", "--nparticles", "16", "--nsamples", "60", "--explicit_backward",
                                         "--outdir", str(tmp_path), "--quiet"])
    assert samples.shape == (60, 3) and np.isfinite(samples).all()
    z = (samples[20:].mean(0) - gp_mean) / np.sqrt(np.diag(gp_cov))
    assert np.abs(z).max() < 1.5
    assert set(np.load(os.path.join(str(tmp_path), "gibbs-eb-16-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}


def test_toy_sb_filter_driver(tmp_path, dev):
    """examples/toy_sb_filter.py (counterpart of experiments/sb/filter.py: one bootstrap filter per sample on the non-separable
    Gaussian Schrodinger bridge, forward observation path from a GP-posterior x0 ('proper') or N(0, I) ('heuristic'))."""
    import os
    mod = _load_example("toy_sb_filter")
    for x0 in ("proper", "heuristic"):
        samples, gp_mean, gp_cov = mod.main(["--d", "3", "--nparticles", "32", "--nsamples", "40", "--x0", x0,
                                             "--outdir", str(tmp_path), "--quiet"])
        assert samples.shape == (40, 3) and np.isfinite(samples).all()
        z = (samples.mean(0) - gp_mean) / np.sqrt(np.diag(gp_cov))
        assert np.abs(z).max() < 1.5, (x0, z)
        assert set(np.load(os.path.join(str(tmp_path), f"filter-{x0}-32-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}


def test_toy_twisted_driver(tmp_path, dev):
    """examples/toy_twisted.py (counterpart of experiments/toy/gp_twisted.py): twisted SMC with the twisting gradient
    through torch autograd; biased but close to the GP posterior, writes the reference's .npz schema."""
    import os
    mod = _load_example("toy_twisted")
    samples, gp_mean, gp_cov = mod.main(["--d", "5", "--nparticles", "64", "--nsamples", "40", "--outdir", str(tmp_path),
                                         "--quiet"])
    assert samples.shape == (40, 5) and np.isfinite(samples).all()
    z = (samples.mean(0) - gp_mean) / np.sqrt(np.diag(gp_cov))
    assert np.abs(z).max() < 1.5
    assert set(np.load(os.path.join(str(tmp_path), "twisted-const-64-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}


def test_toy_csgm_driver(tmp_path, dev):
    """examples/toy_csgm.py (counterpart of experiments/toy/gp_csgm.py, the `csgm` column of the paper's Table 1): the exact
    conditional score integrated by euler_maruyama; the affine drift written out here must be the gradient the reference takes
    by jax.grad -- checked against torch.autograd on the reference's own cond_logpdf --, the samples sit on the GP posterior,
    the .npz schema is the reference's."""
    import os
    mod = _load_example("toy_csgm")
    samples, gp_mean, gp_cov = mod.main(["--d", "6", "--nsamples", "60", "--outdir", str(tmp_path), "--quiet"])
    assert samples.shape == (60, 6) and np.isfinite(samples).all()
    z = (samples.mean(0) - gp_mean) / np.sqrt(np.diag(gp_cov) / 60)
    assert np.abs(z).max() < 4.5, z
    assert set(np.load(os.path.join(str(tmp_path), "csgm-const-666.npz")).files) == {"samples", "gp_mean", "gp_cov"}
    # the written-out gradient against autograd of gp_csgm.py:87-92's log-density (float64, CPU)
    d, F, Q = 4, 0.8, 0.36
    rng = np.random.default_rng(1)
    zs = np.linspace(0., 5., d)
    cov = torch.tensor(np.exp(-np.abs(zs[None, :] - zs[:, None])))
    y0 = torch.tensor(rng.normal(size=d))
    Sx_inv = torch.linalg.inv(F ** 2 * cov + Q * torch.eye(d, dtype=torch.float64))
    M = F * cov @ Sx_inv
    cond_cov = cov + torch.eye(d, dtype=torch.float64) - M @ (F * cov)
    u = torch.tensor(rng.normal(size=d), requires_grad=True)
    lp = torch.distributions.MultivariateNormal(M @ u, covariance_matrix=cond_cov).log_prob(y0)
    want = torch.autograd.grad(lp, u)[0]
    got = M.T @ torch.linalg.inv(cond_cov) @ (y0 - M @ u.detach())
    assert torch.allclose(got, want, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("du,dv,N,T,eb", [(24, 24, 40, 6, True), (33, 17, 255, 4, False), (20, 20, 300, 5, True)])
def test_wide_fused_sweep_explicit_final(du, dv, N, T, eb, oracle, dev):
    """explicit_final=True on the matrix-core path: N + 1 slots, N(0, I) initial particles, initial weights from a
    drift product with the observation pair swapped (gibbs.py:132-138)."""
    toy = toy_gp(du, dv)
    ts = np.linspace(0, 1.0, T + 1)
    br = _bridge(toy, ts, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(du + N)
    x0 = rng.normal(size=du).astype(np.float32)
    bs = rng.integers(0, N + 1, T + 1).astype(np.int32)
    assert br.fused_sweep_supported(N, True)
    sweep = br.sweep_handle(N, eb, True)
    for trial in range(2):
        key = oracle.split(oracle.PRNGKey(21 + trial), 2)[1]
        want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, eb, True, debug=True)
        got = sweep.sweep(key, x0, toy["y0"], bs)
        v = sweep.views()
        _eq(_np(v["us_T"]), want[4], "final particles")
        _eq(_np(v["lw_T"]), want[5], "final log-weights")
        for i, what in enumerate(("x0_next", "us_star_next", "bs_star_next", "acc")):
            _eq(_np(got[i]), want[i], what)
        x0, bs = want[0], want[2]
