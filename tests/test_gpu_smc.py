"""GPU parity of the closure tier beyond gibbs_kernel: csmc_kernel (both backward passes),
bootstrap filter / smoother, pMCMC, pCN, gibbs_init, the SDE simulators -- each against the oracle
on the same keys, bit for bit (the model closures are themselves bit-exact kernels)."""
import numpy as np
import pytest
import torch

from helpers import toy_gp, toy_2d, toy_4d, oracle_model_from

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _eq(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    x = a.view(np.uint32) if a.dtype == np.float32 else a
    y = b.view(np.uint32) if b.dtype == np.float32 else b
    bad = np.flatnonzero(x.ravel() != y.ravel())
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first {bad[:4]}: {a.ravel()[bad[:4]]} vs {b.ravel()[bad[:4]]}"


def _setup(toy, T, Tend, dev):
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    toy = toy()
    ts = np.linspace(0, Tend, T + 1)
    br = fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(-0.5, 1.0), ts, toy["du"],
                                      device=dev)
    return toy, ts, br


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("toy", [toy_2d, toy_4d])
def test_csmc_kernel_both_backward_passes(toy, backward, oracle, dev):
    from fbs_amd.samplers.csmc.csmc import csmc_kernel, forward_pass
    from fbs_amd.samplers.csmc.resamplings import killing
    toy, ts, br = _setup(toy, 20, 1.0, dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(0)
    n = 40
    us_star = rng.normal(size=(21, br.du)).astype(np.float32)
    vs = rng.normal(size=(21, br.dv)).astype(np.float32)
    bs = rng.integers(0, n + 1, 21).astype(np.int32)
    key = oracle.PRNGKey(12)
    us0 = rng.normal(size=(n + 1, br.du)).astype(np.float32)
    lw0 = rng.normal(size=n + 1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    init_sampler = lambda key_, m_: t(us0)
    init_ll = lambda v0, u0s, v1: t(lw0)
    xs, Bs = csmc_kernel(key, t(us_star), bs, t(vs), ts, init_sampler, init_ll, br.transition_sampler,
                         br.transition_logpdf, br.likelihood_logpdf, killing, n, backward=backward)
    wxs, wBs = oracle.csmc_kernel_lg(om, key, us_star, bs, vs, us0, lw0, backward=backward)
    _eq(_np(Bs), wBs, "Bs")
    _eq(_np(xs), wxs, "xs")
    As, lws, uss = forward_pass(oracle.split(key, 2)[0], t(us_star), bs, t(vs), ts, init_sampler, init_ll,
                                br.transition_sampler, br.likelihood_logpdf, killing, n)
    fp = oracle.csmc_forward_pass_lg(om, oracle.split(key, 2)[0], us_star, bs, vs, us0, lw0)
    _eq(_np(As), fp["As"], "As")
    _eq(_np(lws), fp["log_wss"], "log_wss")
    _eq(_np(uss), fp["uss"], "uss")


@pytest.mark.parametrize("resampling", ["stratified", "systematic", "multinomial", "killing"])
def test_bootstrap_filter_smoother_pmcmc_step(resampling, oracle, dev):
    from fbs_amd.samplers import smc
    from fbs_amd.samplers import resampling as R
    toy, ts, br = _setup(toy_2d, 30, 2.0, dev)
    om = oracle_model_from(oracle, br)
    key = oracle.PRNGKey(3)
    k1, k2, k3, k4 = oracle.split(key, 4)
    vs = oracle.lg_fwd_sampler(om, k1, np.array([0.0], np.float32))[::-1].copy()
    n = 128
    init = oracle.normal(k2, (n, 1))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = getattr(R, resampling)
    filt, nell = smc.bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, t(vs), ts,
                                      lambda k, v0, m_: t(init), k3, n, res, log=True, return_last=False)
    wfilt, wnell = oracle.bootstrap_filter_lg(om, k3, vs, init, resampling, return_last=False)
    _eq(_np(filt), wfilt, "filtering samples")
    _eq(np.float32(nell.item()), np.float32(wnell), "nell")
    last, _ = smc.bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, t(vs), ts,
                                   lambda k, v0, m_: t(init), k3, n, res, log=True, return_last=True)
    _eq(_np(last), wfilt[-1], "last samples")
    traj = smc.bootstrap_backward_smoother(k4, filt, t(vs), ts, br.transition_logpdf)
    _eq(_np(traj), oracle.backward_smoother_lg(om, k4, wfilt, vs), "smoother trajectory")
    uT, ell = smc.pmcmc_filter_step(k3, t(vs), t(init), ts, br.transition_sampler, br.likelihood_logpdf, res, n)
    wuT, well = oracle.pmcmc_filter_step_lg(om, k3, vs, init, resampling)
    _eq(_np(uT), wuT, "pmcmc uT")
    _eq(np.float32(ell.item()), np.float32(well), "log_ell")


@pytest.mark.parametrize("delta", [None, 0.1])
def test_pmcmc_kernel(delta, oracle, dev):
    from fbs_amd.samplers import smc
    from fbs_amd.samplers.resampling import stratified
    from fbs_amd.sdes.linear import discretise_linear_sde_np
    toy, ts, br = _setup(toy_2d, 40, 3.0, dev)
    om = oracle_model_from(oracle, br)
    n = 64
    y0 = toy["y0"]
    FQ_T = discretise_linear_sde_np(br.sde, ts[-1], ts[0])
    ref = lambda k, yT, m_: oracle.lg_ref_sampler(toy["m0"], toy["cov0"], FQ_T, 1, k, yT, m_)
    mean_path = (np.asarray(br.sde.mean(ts, ts[0], 1.0), np.float32).reshape(-1, 1) * y0.reshape(1, -1)).astype(np.float32)
    key = oracle.PRNGKey(21)
    ys = oracle.lg_fwd_sampler(om, oracle.PRNGKey(1), y0)
    uT, log_ell = np.array([0.3], np.float32), np.float32(-40.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    state_ys, state_uT, state_ell = ys, uT, log_ell
    for it, k in enumerate(oracle.split(key, 6)):
        got = smc.pmcmc_kernel(k, t(state_uT), float(state_ell), t(state_ys), t(y0), ts, br.fwd_ys_sampler, br.sde,
                               br.ref_sampler, br.transition_sampler, br.likelihood_logpdf, stratified, n, delta=delta)
        want = oracle.pmcmc_kernel_lg(om, k, state_uT, state_ell, state_ys, y0, n, ref, mean_path, delta)
        _eq(_np(got[0]).reshape(-1), np.asarray(want[0]).reshape(-1), f"uT it{it}")
        _eq(np.float32(got[1].item()), np.float32(want[1]), f"log_ell it{it}")
        _eq(_np(got[2]), want[2], f"ys it{it}")
        assert bool(got[3].is_accepted.item()) == want[3]
        state_uT, state_ell, state_ys = np.asarray(want[0]).reshape(-1), want[1], want[2]


def test_gibbs_init_filter_and_smoother_run(oracle, dev):
    """gibbs_init (gibbs.py:23-65): both methods produce a valid (x0, us_star) on the device; the
    pieces it composes are parity-tested above, here the key routing (six-way split, the init_sampler
    that ignores its arguments and reuses key_u0) is checked against the oracle primitives."""
    from fbs_amd.samplers import gibbs_init
    toy, ts, br = _setup(toy_2d, 25, 2.0, dev)
    om = oracle_model_from(oracle, br)
    key = oracle.PRNGKey(31)
    y0 = torch.from_numpy(toy["y0"]).to(dev)
    n = 50
    x0, us_star = gibbs_init(key, y0, (1,), ts, br.fwd_sampler, br.sde, br.unpack, br.transition_sampler,
                             br.transition_logpdf, br.likelihood_logpdf, n, method='filter', marg_y=False)
    k_fwd, k_bridge, k_u0, k_bf, k_fwd2, k_bwd = oracle.split(key, 6)
    path = oracle.lg_fwd_sampler(om, k_fwd, np.array([0.0, toy["y0"][0]], np.float32))
    vs = path[::-1, 1:].copy()
    init = oracle.normal(k_u0, (n, 1))
    last, _ = oracle.bootstrap_filter_lg(om, k_bf, vs, init, "stratified", return_last=True)
    _eq(_np(x0), last[0], "approx_x0")
    want_us = oracle.lg_fwd_sampler(om, k_fwd2, np.array([last[0, 0], toy["y0"][0]], np.float32))[::-1, :1]
    _eq(_np(us_star), want_us.copy(), "approx_us_star")
    x0s, us_s = gibbs_init(key, y0, (1,), ts, br.fwd_sampler, br.sde, br.unpack, br.transition_sampler,
                           br.transition_logpdf, br.likelihood_logpdf, n, method='smoother', marg_y=False)
    filt, _ = oracle.bootstrap_filter_lg(om, k_bf, vs, init, "stratified", return_last=False)
    _eq(_np(x0s), filt[-1, 0], "smoother x0")
    _eq(_np(us_s), oracle.backward_smoother_lg(om, k_bwd, filt, vs), "smoother us_star")
    with pytest.raises(ValueError):
        gibbs_init(key, y0, (1,), ts, br.fwd_sampler, br.sde, br.unpack, br.transition_sampler,
                   br.transition_logpdf, br.likelihood_logpdf, n, method='nope', marg_y=False)


def test_doob_bridge_and_marg_y_gibbs(oracle, dev):
    from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE, doob_bridge_simulator
    from fbs_amd.sdes.linear import _bridge_drift_coeffs
    for sde in (StationaryConstLinearSDE(-0.5, 1.0), StationaryLinLinearSDE(0.02, 4.0, 0.0, 1.0)):
        ts = np.linspace(0, 1, 11)
        T, nsub = 10, 20
        key = oracle.PRNGKey(4)
        x0 = np.array([0.5, -1.0, 2.0], np.float32)
        xT = np.array([5.0, 0.0, -3.0], np.float32)
        got = doob_bridge_simulator(key, sde, torch.from_numpy(x0).to(dev), torch.from_numpy(xT).to(dev), ts,
                                    integration_nsteps=nsub, replace=True)
        A, B, S, ddt = np.zeros(T * nsub), np.zeros(T * nsub), np.zeros(T * nsub), np.zeros(T)
        for k in range(T):
            h = abs(ts[k + 1] - ts[k]) / nsub
            ddt[k] = h
            for j, t_ in enumerate(np.linspace(ts[k], ts[k + 1] - h, nsub)):
                A[k * nsub + j], B[k * nsub + j] = _bridge_drift_coeffs(sde, float(t_), float(ts[-1]))
                S[k * nsub + j] = float(sde.dispersion(float(t_)))
        want = oracle.doob_bridge_np(key, A, B, S, ddt, x0, xT, T, nsub, True)
        _eq(_np(got), want, "doob bridge path")
        # the bridge hits its target (tests/test_sdes.py:115-132 restated): before replacement
        free = _np(doob_bridge_simulator(key, sde, torch.from_numpy(x0).to(dev), torch.from_numpy(xT).to(dev),
                                         np.linspace(0, 1, 101), integration_nsteps=100, replace=False))
        np.testing.assert_allclose(free[-1], xT, rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("toy,n,T", [(toy_2d, 20, 12), (toy_4d, 64, 9), (toy_2d, 1000, 5)])
def test_marg_y_gibbs_sweep_equals_oracle(toy, n, T, oracle, dev):
    """gibbs_kernel(marg_y=True): the observation path is re-drawn by bridge_sampler = doob_bridge_simulator with 100
    sub-steps (gibbs.py:17-20,130), then the usual conditional-SMC sweep -- the whole sweep against the oracle."""
    from fbs_amd.samplers import gibbs_kernel
    from fbs_amd.sdes.linear import _bridge_drift_coeffs
    toy, ts, br = _setup(toy, T, 1.0, dev)
    om = oracle_model_from(oracle, br)
    nsub = 100
    A, B, S, ddt = np.zeros(T * nsub), np.zeros(T * nsub), np.zeros(T * nsub), np.zeros(T)
    for k in range(T):
        h = abs(ts[k + 1] - ts[k]) / nsub
        ddt[k] = h
        for j, t_ in enumerate(np.linspace(ts[k], ts[k + 1] - h, nsub)):
            A[k * nsub + j], B[k * nsub + j] = _bridge_drift_coeffs(br.sde, float(t_), float(ts[-1]))
            S[k * nsub + j] = float(br.sde.dispersion(float(t_)))
    bridge = lambda key_, y_first, y_last: oracle.doob_bridge_np(key_, A, B, S, ddt, y_first, y_last, T, nsub, True)
    rng = np.random.default_rng(8)
    x0 = rng.normal(size=br.du).astype(np.float32)
    bs = rng.integers(0, n, T + 1).astype(np.int32)
    key = oracle.PRNGKey(2)
    got = gibbs_kernel(key, torch.from_numpy(x0).to(dev), torch.from_numpy(toy["y0"]).to(dev), None, bs, ts,
                       br.fwd_sampler, br.sde, br.unpack, n, br.transition_sampler, br.transition_logpdf,
                       br.likelihood_logpdf, marg_y=True)
    want = oracle.gibbs_kernel_lg_marg_y(om, key, x0, toy["y0"], bs, n, bridge)
    for a, b, w in zip(got, want, ("x0", "us_star", "bs_star", "acc")):
        _eq(_np(a), b, w)


def test_discrete_time_simulator_equals_oracle(oracle, dev):
    """fbs/sdes/simulators.py:109-123."""
    from fbs_amd.sdes import discrete_time_simulator
    ts = np.linspace(0, 1.5, 14)
    key = oracle.PRNGKey(21)
    x0 = np.array([[0.5, -1.0, 0.25], [2.0, 0.1, -0.75]], np.float32)
    # closures made of single float32 operations, so that torch and numpy round identically
    f_t = lambda x, t1, t0: x * float(np.float32(0.9)) + float(np.float32(0.1 * (t1 - t0)))
    f_n = lambda x, t1, t0: ((x * np.float32(0.9)).astype(np.float32) + np.float32(0.1 * (t1 - t0))).astype(np.float32)
    q = lambda t1, t0: 0.3 + 0.5 * (t1 - t0)
    got = discrete_time_simulator(key, torch.from_numpy(x0).to(dev), ts, f_t, q)
    want = oracle.discrete_time_simulator_np(key, x0, ts, f_n, q)
    _eq(_np(got), want, "discrete_time_simulator")


def test_euler_maruyama_and_reverse_simulator(oracle, dev):
    from fbs_amd.sdes import euler_maruyama, reverse_simulator
    ts = np.linspace(0, 1, 9)
    key = oracle.PRNGKey(6)
    x0 = np.array([[0.5, -1.0], [2.0, 0.1]], np.float32)
    drift = lambda x, t: -0.5 * x * (1.0 + t)
    disp = lambda t: 1.0 + 0.5 * t
    got = euler_maruyama(key, torch.from_numpy(x0).to(dev), ts, drift, disp, integration_nsteps=3, return_path=True)
    want = oracle.euler_maruyama_np(key, x0, ts, drift, disp, integration_nsteps=3, return_path=True)
    # the update (x + f * ddt) + c * xi is one kernel with the noise drawn inside (fbsmi_em_update, no contraction): given the
    # same drift values the path is the oracle's bit for bit (the drift here is two float32 multiplications on either side)
    _eq(_np(got), want, "euler_maruyama path")
    term = euler_maruyama(key, torch.from_numpy(x0).to(dev), ts, drift, disp, integration_nsteps=3)
    _eq(_np(term), want[-1], "euler_maruyama terminal value")
    # a ragged size (not a multiple of four, beyond one workgroup) and a scalar-valued drift closure
    x1 = oracle.normal(oracle.PRNGKey(11), (1027,))
    got = euler_maruyama(key, torch.from_numpy(x1).to(dev), ts, lambda x, t: 0.25, disp, integration_nsteps=2)
    want = oracle.euler_maruyama_np(key, x1, ts, lambda x, t: np.float32(0.25), disp, integration_nsteps=2)
    _eq(_np(got), want, "euler_maruyama ragged")
    # reverse_simulator keeps N(0,1) stationary for the OU process (tests/test_sdes.py:166-194, loose)
    u0 = oracle.normal(oracle.PRNGKey(9), (20000,))
    out = reverse_simulator(oracle.PRNGKey(10), torch.from_numpy(u0).to(dev), np.linspace(0, 1, 65),
                            lambda u, t: -u, lambda u, t: -0.5 * u, lambda t: 1.0, integration_nsteps=1)
    assert abs(float(out.mean())) < 0.05 and abs(float(out.var()) - 1.0) < 0.08
    with pytest.raises(NotImplementedError):
        reverse_simulator(key, torch.from_numpy(u0).to(dev), ts, None, None, None, integrator='rk4')


def test_twisted_smc_runs(oracle, dev):
    """twisted_smc (smc.py:261-309) on a conjugate Gaussian toy: weights normalised, particles finite."""
    from fbs_amd.samplers import twisted_smc
    from fbs_amd.samplers.resampling import stratified
    from fbs_amd import ops
    ts = np.linspace(0, 1, 11)
    n = 256
    y = torch.tensor([0.7], device=dev)

    def init_sampler(k, m_):
        return ops.normal(k, (m_, 1), device=dev)

    def logn(x, mu, sd):
        return (-0.5 * ((x - mu) / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum(-1)

    transition_logpdf = lambda xs, xp, t: logn(xs, 0.9 * xp, 0.3)
    twisting_logpdf = lambda y_, xs, t: logn(y_, xs, 1.0)
    prop_sampler = lambda k, xp, t, y_: 0.9 * xp + 0.3 * ops.normal(k, tuple(xp.shape), device=dev)
    prop_logpdf = lambda xs, xp, t, y_: logn(xs, 0.9 * xp, 0.3)
    xs, lw = twisted_smc(oracle.PRNGKey(1), y, ts, init_sampler, transition_logpdf, twisting_logpdf, prop_sampler,
                         prop_logpdf, stratified, n)
    assert xs.shape == (n, 1) and torch.isfinite(xs).all()
    assert abs(float(torch.exp(lw).sum()) - 1.0) < 1e-4


def test_twisted_smc_equals_oracle(oracle, dev):
    """twisted_smc (fbs/samplers/smc.py:261-309) against its numpy restatement: resampling indices bit for bit, final
    particles and weights bit for bit.  The closures are Gaussian log-densities spelled as single float32 operations
    (no transcendental inside), so torch and numpy round identically and nothing but the sampler is compared."""
    from fbs_amd import ops
    from fbs_amd.samplers import stratified
    from fbs_amd.samplers.smc import twisted_smc
    n, T = 300, 11
    ts = np.linspace(0, 1, T + 1)
    y = np.float32(0.7)
    f32 = np.float32

    def quad(a, b, c2):          # -(c2 (a - b)^2) / 2: subtract, multiply, multiply, multiply -- one rounding each.  No
        d = a - b                # division: torch divides by a scalar through its reciprocal, numpy does not
        return ((d * d) * c2) * -0.5

    # torch side
    yt = torch.tensor(float(y), device=dev)
    tl_t = lambda xs, xp, t: quad(xs, xp * 0.95, 25.0).reshape(-1)
    tw_t = lambda y_, xs, t: quad(xs, y_ * (0.5 + 0.25 * float(t)), 2.0).reshape(-1)
    ps_t = lambda k, xp, t, y_: xp * 0.9 + ops.normal(k, tuple(xp.shape), device=dev) * 0.25
    pl_t = lambda xs, xp, t, y_: quad(xs, xp * 0.9, 16.0).reshape(-1)
    init_t = lambda k, m: ops.normal(k, (m, 1), device=dev)
    # numpy side (the same operations)
    qn = lambda a, b, c2: ((((a - b).astype(f32) * (a - b).astype(f32)).astype(f32) * f32(c2)).astype(f32) * f32(-0.5)).astype(f32)
    tl_n = lambda xs, xp, t: qn(xs, (xp * f32(0.95)).astype(f32), 25.0).reshape(-1)
    tw_n = lambda y_, xs, t: qn(xs, f32(y_ * f32(0.5 + 0.25 * float(t))), 2.0).reshape(-1)
    ps_n = lambda k, xp, t, y_: ((xp * f32(0.9)).astype(f32) + (oracle.normal(k, xp.shape) * f32(0.25)).astype(f32)).astype(f32)
    pl_n = lambda xs, xp, t, y_: qn(xs, (xp * f32(0.9)).astype(f32), 16.0).reshape(-1)
    init_n = lambda k, m: oracle.normal(k, (m, 1))
    key = oracle.PRNGKey(1)
    xs, lw = twisted_smc(key, yt, ts, init_t, tl_t, tw_t, ps_t, pl_t, stratified, n)
    wxs, wlw, winds = oracle.twisted_smc_np(key, y, ts, init_n, tl_n, tw_n, ps_n, pl_n, oracle.stratified, n)
    _eq(_np(xs), wxs, "twisted_smc particles")
    _eq(_np(lw), wlw, "twisted_smc log-weights")
    assert abs(float(torch.exp(lw).sum()) - 1.0) < 1e-4 and len(winds) == T


def test_sharded_module_world1_on_gpu(oracle, dev):
    """fbs_amd.sharded with the real (libfbsmi) backend, one rank: same answer as the oracle; and
    the sliced PRNG draws equal slices of the full draw."""
    from fbs_amd import sharded, ops
    toy, ts, br = _setup(toy_4d, 15, 1.0, dev)
    om = oracle_model_from(oracle, br)
    N = 96
    rng = np.random.default_rng(5)
    x0 = rng.normal(size=br.du).astype(np.float32)
    bs = rng.integers(0, N, 16).astype(np.int32)
    key = oracle.PRNGKey(77)
    sh = sharded.ParticleShards(N)
    out = sharded.gibbs_kernel(key, torch.from_numpy(x0).to(dev), torch.from_numpy(toy["y0"]).to(dev), None, bs, ts,
                               br.fwd_sampler, br.sde, br.unpack, N, br.transition_sampler, br.transition_logpdf,
                               br.likelihood_logpdf, sh)
    want = oracle.gibbs_kernel_lg(om, key, x0, toy["y0"], bs, N, True, False)
    for a, b, w in zip(out, want, ("x0", "us_star", "bs_star", "acc")):
        _eq(_np(a), b, w)
    full = _np(ops.normal(key, (N, 3), device=dev))
    part = _np(ops.normal(key, (N, 3), device=dev, rows=(32, 40)))
    _eq(part, full[32:72], "normal rows slice")
    fullu = _np(ops.uniform(key, (N,), device=dev))
    _eq(_np(ops.uniform(key, (N,), device=dev, rows=(90, 6))), fullu[90:96], "uniform rows slice")
    # a row-sliced transition equals the slice of the full transition
    us_prev = torch.from_numpy(rng.normal(size=(N, br.du)).astype(np.float32)).to(dev)
    v_prev = torch.from_numpy(rng.normal(size=br.dv).astype(np.float32)).to(dev)
    whole = _np(br.transition_sampler(us_prev, v_prev, ts[3], key))
    piece = _np(br.transition_sampler(us_prev[24:72].contiguous(), v_prev, ts[3], key, row_slice=(24, 48, N)))
    _eq(piece, whole[24:72], "row-sliced transition")


@pytest.mark.parametrize("resampling", ["stratified", "systematic"])
@pytest.mark.parametrize("toy,n,T", [(toy_2d, 128, 30), (toy_4d, 1000, 12), (toy_2d, 70000, 6),
                                     (toy_2d, 10, 20), (toy_4d, 256, 8), (toy_4d, 257, 6),   # around the one-launch threshold
                                     (lambda: toy_gp(100), 100, 8),      # the reference's gp_filter / gp_pmcmc scale
                                     (lambda: toy_gp(33, 17), 37, 6),    # wide, odd sizes, tiles straddling du
                                     (lambda: toy_gp(20), 256, 5),
                                     (lambda: toy_gp(20), 300, 5), (lambda: toy_gp(33, 17), 1000, 4),    # several tiles
                                     # powers of two with 2..256 tiles: the TWO-launch filter step (tree-walking searches)
                                     (toy_4d, 512, 6), (toy_2d, 4096, 8), (toy_2d, 65536, 3)])
def test_fused_filters_match_oracle(toy, n, T, resampling, oracle, dev):
    """The fused (hipGraph) bootstrap_filter / pmcmc_filter_step of the analytic model, reached through
    the unchanged fbs_amd.samplers.smc signatures, against the oracle -- and against the closure tier."""
    from fbs_amd.samplers import smc
    from fbs_amd.samplers import resampling as R
    toy, ts, br = _setup(toy, T, 2.0, dev)
    om = oracle_model_from(oracle, br)
    k1, k2, k3 = oracle.split(oracle.PRNGKey(41), 3)
    vs = oracle.lg_fwd_sampler(om, k1, toy["y0"])[::-1].copy()
    init = oracle.normal(k2, (n, br.du))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = getattr(R, resampling)
    init_sampler = lambda k, v0, m_: t(init)
    filt, nell = smc.bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, t(vs), ts, init_sampler, k3, n, res,
                                      log=True, return_last=False)
    wfilt, wnell = oracle.bootstrap_filter_lg(om, k3, vs, init, resampling, return_last=False)
    _eq(_np(filt), wfilt, "fused filtering path")
    _eq(np.float32(nell.item()), np.float32(wnell), "fused nell")
    last, nell2 = smc.bootstrap_filter(br.transition_sampler, br.likelihood_logpdf, t(vs), ts, init_sampler, k3, n, res)
    _eq(_np(last), wfilt[-1], "fused last")
    _eq(np.float32(nell2.item()), np.float32(wnell), "fused nell (return_last)")
    uT, ell = smc.pmcmc_filter_step(k3, t(vs), t(init), ts, br.transition_sampler, br.likelihood_logpdf, res, n)
    wuT, well = oracle.pmcmc_filter_step_lg(om, k3, vs, init, resampling)
    _eq(_np(uT), wuT, "fused pmcmc uT")
    _eq(np.float32(ell.item()), np.float32(well), "fused log_ell")
    # the closure tier (plain lambdas hide the descriptor) gives the same bits
    if n <= 1000:
        uT2, ell2 = smc.pmcmc_filter_step(k3, t(vs), t(init), ts, lambda *a: br.transition_sampler(*a),
                                          lambda *a: br.likelihood_logpdf(*a), res, n)
        _eq(_np(uT2), wuT, "closure-tier pmcmc uT")
        _eq(np.float32(ell2.item()), np.float32(well), "closure-tier log_ell")
