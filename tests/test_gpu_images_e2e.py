"""GPU: the closure tier with a (randomly initialised) UNet score on an MNIST-shaped inpainting task
(BASELINE config 3 in miniature): gibbs_kernel / pmcmc_kernel run end to end with the mask threaded
through **kwargs exactly as experiments/imgs/inpainting.py does, the network is evaluated once per
step (cache), and sharded and unsharded ensembles agree."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, dim=8, T=6):
    from fbs_amd.images import ImageRestore
    from fbs_amd.score import ScoreBridge
    from fbs_amd.sdes import StationaryLinLinearSDE
    from fbs_amd.unet import UNet
    torch.manual_seed(0)
    Tend = 2.0
    ts = np.linspace(0, Tend, T + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=Tend)      # inpainting.py:78
    ds = ImageRestore("inpaint-15", (28, 28, 1), device=dev)
    net = UNet(dt=Tend / 200, dim=dim, in_channels=1, upsampling="pixel_shuffle").to(dev).eval()
    calls = {"n": 0}

    def score_fn(x, t):
        calls["n"] += 1
        with torch.no_grad():
            return net(x, t).reshape(x.shape)

    return ds, ScoreBridge(score_fn, ds, sde, ts, chunk=64), sde, ts, calls


def test_gibbs_and_pmcmc_with_unet_score(oracle, dev):
    from fbs_amd import ops
    from fbs_amd.samplers import gibbs_kernel, pmcmc_kernel, stratified
    ds, sb, sde, ts, calls = _setup(dev)
    T = len(ts) - 1
    key = oracle.PRNGKey(996)                                                   # imgs_gibbs.sh:37 seed
    key, k_img, k_mask = oracle.split(key, 3)
    img = ops.uniform(k_img, (28, 28, 1), device=dev)
    mask = ds.gen_mask(k_mask)
    x_true, y0 = ds.unpack(img, mask)
    assert x_true.shape == (225, 1) and y0.shape == (559, 1)                    # du=225, dv=559
    N = 24
    x0 = torch.zeros(225, 1, device=dev)
    bs = np.zeros(T + 1, np.int32)
    for eb, ef in ((True, False), (True, True), (False, False)):
        calls["n"] = 0
        out = gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, N, sb.transition_sampler,
                           sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False, explicit_backward=eb,
                           explicit_final=ef, mask_=mask)
        x0n, usn, bsn, acc = out
        assert x0n.shape == (225, 1) and usn.shape == (T + 1, 225, 1) and bsn.shape == (T + 1,)
        assert torch.isfinite(usn).all() and acc.dtype == torch.bool
        rows = N + 1 if ef else N
        per_step = -(-rows // 64)
        # one network evaluation per SMC step (+1 for the explicit-final initial weights), not two
        assert calls["n"] == per_step * (T + (1 if ef else 0)), calls["n"]
    # pMCMC over the same closures
    ys = sb.fwd_ys_sampler(oracle.PRNGKey(1), y0)
    uT, ell, ys2, state = pmcmc_kernel(oracle.PRNGKey(2), x0, -1e9, ys, y0, ts, sb.fwd_ys_sampler, sde,
                                       sb.ref_sampler, sb.transition_sampler, sb.likelihood_logpdf, stratified, N,
                                       delta=0.005, mask_=mask)
    assert uT.shape == (225, 1) and torch.isfinite(ell) and bool(state.is_accepted.item())


def test_sharded_world1_equals_unsharded_with_unet(oracle, dev):
    from fbs_amd import ops, sharded
    from fbs_amd.samplers import gibbs_kernel
    ds, sb, sde, ts, _ = _setup(dev)
    T = len(ts) - 1
    key = oracle.PRNGKey(5)
    mask = ds.gen_mask(oracle.PRNGKey(6))
    img = ops.uniform(oracle.PRNGKey(7), (28, 28, 1), device=dev)
    _, y0 = ds.unpack(img, mask)
    x0 = torch.zeros(225, 1, device=dev)
    bs = np.arange(T + 1, dtype=np.int32) % 16
    a = gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, 16, sb.transition_sampler,
                     sb.transition_logpdf, sb.likelihood_logpdf, mask_=mask)
    b = sharded.gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, 16, sb.transition_sampler,
                             sb.transition_logpdf, sb.likelihood_logpdf, sharded.ParticleShards(16), mask_=mask)
    if all(torch.equal(u, v) for u, v in zip(a, b)):
        return
    # Bit equality is a property of the sampler, not of MIOpen: if the network's own convolutions are not
    # reproducible run to run on this box (some float32 solvers accumulate with atomics), the two runs may differ
    # in the last bits of the network output and nothing more can be asked than closeness before the first
    # resampling decision flips.
    again = gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, 16, sb.transition_sampler,
                         sb.transition_logpdf, sb.likelihood_logpdf, mask_=mask)
    network_reproducible = all(torch.equal(u, v) for u, v in zip(a, again))
    assert not network_reproducible, "sharded (world 1) and unsharded sweeps differ although the network is reproducible"


@pytest.mark.parametrize("argv", [
    ["--task", "inpaint", "--rect_size", "15", "--method", "gibbs-eb-ef"],
    ["--task", "supr", "--rate", "4", "--method", "filter"],
    ["--task", "inpaint", "--rect_size", "8", "--method", "pmcmc-0.005"],
    ["--task", "supr", "--rate", "4", "--sb", "--method", "gibbs"],
    ["--task", "inpaint", "--rect_size", "15", "--method", "twisted"],
    ["--task", "supr", "--rate", "2", "--method", "csgm"],
])
def test_image_drivers_run_end_to_end(argv, tmp_path, dev):
    """examples/imgs_restore.py: the counterparts of experiments/imgs/inpainting.py, supr.py and experiments/sb_imgs/supr.py
    (same flags and result files), at toy sizes, with a small randomly initialised UNet."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import imgs_restore
    out = imgs_restore.main(argv + ["--dim", "8", "--nparticles", "12", "--nsamples", "2", "--ny0s", "1", "--test_nsteps", "5",
                                    "--chunk", "8", "--fp32", "--quiet", "--outdir", str(tmp_path)])
    assert out.shape == (2, 28, 28, 1) and np.isfinite(out).all()
    files = sorted(os.listdir(tmp_path))
    assert any(f.endswith("-true.npz") for f in files) and any(("gibbs" in f or "filter" in f or "pmcmc" in f or "twisted" in f or "csgm" in f) and f.endswith(".npy") for f in files)


def test_image_twisted_closures_closed_form_and_driver_pieces(oracle, dev):
    """fbs_amd/twisted.py (experiments/imgs/inpainting_twisted.py:97-154) with a LINEAR stand-in score s(x, t) = -c x, for which
    everything has a closed form: the one-step denoising estimate is kappa x with kappa = 1 + dt (0.5 beta + beta (-c)), so the
    gradient of the twisting log-density that autograd sends through the score is kappa (y - kappa x) / s^2 on the observed
    pixels and 0 elsewhere; the proposal's mean, its log-density and the weights of twisted_smc follow.  Then one run of
    twisted_smc itself (stratified resampling: libfbsmi) on whole-image particles."""
    import math
    from fbs_amd import ops
    from fbs_amd.images import ImageRestore
    from fbs_amd.samplers import stratified
    from fbs_amd.sdes import StationaryLinLinearSDE, make_linear_sde
    from fbs_amd.twisted import make_image_twisted
    Tend, T, n, c = 2.0, 8, 16, 0.7
    ts = np.linspace(0, Tend, T + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=Tend)
    ds = ImageRestore("inpaint-5", (12, 12, 2), device=dev)
    mask = ds.gen_mask(oracle.PRNGKey(4))
    tw = make_image_twisted(lambda x, t: -c * x, ds, sde, ts, n)
    rng = np.random.default_rng(0)
    uv = torch.from_numpy(rng.normal(size=(n, 12, 12, 2)).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.normal(size=tuple(ds.unpack(uv[0], mask)[1].shape)).astype(np.float32)).to(dev)
    t = float(ts[3])
    dt = Tend / T
    beta = sde.beta(Tend - t)
    kappa = 1.0 + dt * (0.5 * beta - beta * c)
    F, Q = make_linear_sde(sde)[0](Tend - t, 0.0)
    s2 = float(F) ** 2 * 0.06 + float(Q)
    grad = torch.zeros_like(uv).reshape(n, 144, 2)
    obs = ds.unpack(uv, mask)[1]
    grad[:, mask.obs_inds_ravelled] = kappa * (y.unsqueeze(0) - kappa * obs) / s2
    want_cd = (0.5 * beta - beta * c) * uv + beta * grad.reshape(uv.shape)
    got_cd = tw.reverse_cond_drift(uv, t, y, mask)
    assert torch.allclose(got_cd, want_cd, rtol=2e-5, atol=2e-5), (got_cd - want_cd).abs().max().item()
    want_tw = (-0.5 * (y.unsqueeze(0) - kappa * obs) ** 2 / s2 - 0.5 * math.log(2 * math.pi * s2)).reshape(n, -1).sum(1)
    assert torch.allclose(tw.twisting_logpdf(y, uv, t, mask_=mask), want_tw, rtol=2e-5, atol=1e-3)
    # proposal: mean + sqrt(dt) b z with z = jax.random.normal(key, (n, w, h, c)) drawn by libfbsmi
    key = oracle.PRNGKey(9)
    z = torch.from_numpy(oracle.normal(key, (n, 12, 12, 2))).to(dev)
    want_prop = (uv + want_cd * dt) + math.sqrt(dt) * math.sqrt(beta) * z
    got_prop = tw.twisting_prop_sampler(key, uv, t, y, mask_=mask)
    assert torch.allclose(got_prop, want_prop, rtol=2e-5, atol=2e-5)
    lp = tw.twisting_prop_logpdf(got_prop, uv, t, y, mask_=mask)
    want_lp = (-0.5 * z.double() ** 2 - 0.5 * math.log(2 * math.pi * dt * beta)).reshape(n, -1).sum(1)
    assert torch.allclose(lp.double(), want_lp, rtol=1e-4, atol=5e-2)
    # the whole filter + the final categorical draw
    out = tw.conditional_sampler(oracle.PRNGKey(11), y, stratified, mask_=mask)
    assert out.shape == (12, 12, 2) and torch.isfinite(out).all()


def test_image_csgm_sampler_closed_form(oracle, dev):
    """fbs_amd/csgm.py (experiments/imgs/inpainting_csgm.py:88-121) against a float64 replay of the same recursion with a stand-in
    score that couples the pixels, s(x, t) = -c x + g mean(x): the initial draw, every step's re-noised observation (its own
    key), the drift through concat / unpack and the Euler-Maruyama noise (the step's slice of ONE draw of (nsteps, *x_shape)) --
    the reference's key schedule, draws from the oracle's generator."""
    import math
    from fbs_amd.csgm import make_image_csgm
    from fbs_amd.images import ImageRestore
    from fbs_amd.sdes import StationaryLinLinearSDE, make_linear_sde
    Tend, T, c, g = 2.0, 9, 0.6, 0.8
    ts = np.linspace(0, Tend, T + 1)
    dt = Tend / T
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=Tend)
    ds = ImageRestore("inpaint-5", (12, 12, 2), device=dev)
    mask = ds.gen_mask(oracle.PRNGKey(4))
    img = torch.from_numpy(np.random.default_rng(1).uniform(size=(12, 12, 2)).astype(np.float32)).to(dev)
    y0 = ds.unpack(img, mask)[1]
    sampler = make_image_csgm(lambda x, t: -c * x + g * x.mean(), ds, sde, ts)
    key = oracle.PRNGKey(21)
    got = sampler(key, y0, mask)
    # float64 replay
    discretise = make_linear_sde(sde)[0]
    y = y0.cpu().numpy().astype(np.float64)
    key_init, key_sde = oracle.split(key, 2)
    u = oracle.normal(key_init, tuple(ds.unobs_shape)).astype(np.float64)
    key_scan, key_est = oracle.split(key_sde, 2)
    key_ests = oracle.split(key_est, T)
    rnds = oracle.normal(key_scan, (T,) + tuple(ds.unobs_shape)).astype(np.float64)
    for k in range(T):
        s_ = Tend - ts[k]
        beta = sde.beta(s_)
        F, Q = (float(x) for x in discretise(s_, 0.0))
        v_hat = F * y + math.sqrt(Q) * oracle.normal(key_ests[k], y.shape).astype(np.float64)
        mean = (u.sum() + v_hat.sum()) / (u.size + v_hat.size)
        u = u + (0.5 * beta * u + beta * (-c * u + g * mean)) * dt + math.sqrt(beta) * math.sqrt(dt) * rnds[k]
    assert got.shape == tuple(ds.unobs_shape)
    np.testing.assert_allclose(got.cpu().numpy(), u, rtol=5e-5, atol=5e-5)
