"""GPU parity of the fused score-network SMC step (fbsmi_em_concat / fbsmi_em_finish /
fbsmi_em_transition_logpdf, fbs_amd/csrc/fbsmi_em.hip) against the numpy restatement oracle/em.py of
experiments/imgs/inpainting.py:102-147 -- bit for bit: particle rows, log-weights, and through them the
ancestors of a whole closure-tier sweep.  Shapes: MNIST inpaint-15 (du=225, dv=559: BASELINE config 3),
MNIST supr-4 (du=735, dv=49: config 4), CelebA-64 inpaint-32 (du=3072, dv=9216: config 5, at the
per-GPU shard size N=2048), plus ragged ones."""
import ctypes as C

import numpy as np
import pytest
import torch

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


def _mask(task, shape, dev, key):
    from fbs_amd.images import ImageRestore
    from fbs_amd.score import EMMask
    ds = ImageRestore(task, shape, sr_random=(key is not None), device=dev)
    mask = ds.gen_mask(key if key is not None else np.array([0, 1], np.uint32))
    return ds, mask, EMMask(mask, shape[2], dev)


def _tables(O, mask, c):
    from oracle import em
    return em.element_tables(_np(mask.unobs_inds_ravelled), _np(mask.obs_inds_ravelled), c)


CASES = [  # task, image shape, rows
    ("inpaint-15", (28, 28, 1), 37),
    ("supr-4", (28, 28, 1), 64),
    ("inpaint-8", (16, 16, 3), 130),
    ("inpaint-5", (12, 12, 3), 9),
    ("supr-2", (8, 8, 2), 33),
]


@pytest.mark.parametrize("task,shape,n", CASES)
@pytest.mark.parametrize("out_dtype", ["f32", "bf16"])
def test_concat_matches_oracle(task, shape, n, out_dtype, oracle, dev):
    from fbs_amd import _lib
    from oracle import em
    ds, mask, emk = _mask(task, shape, dev, oracle.PRNGKey(3))
    u_off, v_off, role = _tables(oracle, mask, shape[2])
    _eq(_np(emk.u_off), u_off, "u_off"); _eq(_np(emk.v_off), v_off, "v_off"); _eq(_np(emk.role), role, "role")
    rng = np.random.default_rng(1)
    us = rng.normal(size=(n + 3, emk.du)).astype(np.float32)
    vp = rng.normal(size=emk.dv).astype(np.float32)
    for A in (None, rng.integers(0, n + 3, n).astype(np.int32)):
        rows = n + 3 if A is None else n
        want = em.concat(us, A, vp, role)
        tdt = torch.float32 if out_dtype == "f32" else torch.bfloat16
        img = torch.empty((rows, emk.D), dtype=tdt, device=dev)
        ust, vpt = torch.from_numpy(us).to(dev), torch.from_numpy(vp).to(dev)
        At = torch.from_numpy(A).to(dev) if A is not None else None
        _lib.call("fbsmi_em_concat", emk.ref, ust.data_ptr(), At.data_ptr() if A is not None else None, vpt.data_ptr(),
                  rows, 0 if out_dtype == "f32" else 1, img.data_ptr(), 0)
        torch.cuda.synchronize()
        if out_dtype == "f32":
            _eq(_np(img), want, "img")
            # the product's own concat (fbs/data/images.py:355-363 restated with torch ops) agrees
            src = ust if A is None else ust[At.long()]
            ref = ds.concat(src.reshape(rows, -1, shape[2]), vpt.reshape(-1, shape[2]), mask)
            _eq(_np(ref.reshape(rows, -1)), want, "ImageRestore.concat")
        else:
            _eq(_np(img.view(torch.int16)).view(np.uint16), em.to_bf16_bits(want), "img bf16")


def _finish(emk, us, A, net, mode, cx, cs, dt, sd, v, vp, key, n_total, row0, pin_row, pin_val, dev, want_us=True,
            want_lw=True, net_A=None):
    from fbs_amd import _lib
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
    ust, At, vt, vpt, pt, nAt = t(us), t(A), t(v), t(vp), t(pin_val), t(net_A)
    if isinstance(net, torch.Tensor):
        nt = net
    else:
        nt = t(net)
    n = (net_A.size if net_A is not None else nt.shape[0])
    out = torch.full((n, emk.du), np.nan, dtype=torch.float32, device=dev) if want_us else None
    lw = torch.full((n,), np.nan, dtype=torch.float32, device=dev) if want_lw else None
    p = lambda x: x.data_ptr() if x is not None else None
    _lib.call("fbsmi_em_finish", emk.ref, p(ust), p(At), nt.data_ptr(), p(nAt), 0 if nt.dtype == torch.float32 else 1,
              mode, cx, cs, dt, sd, p(vt), p(vpt), int(key[0]), int(key[1]), n_total, row0, n, pin_row, p(pt), p(out),
              p(lw), 0)
    torch.cuda.synchronize()
    return (_np(out) if want_us else None), (_np(lw) if want_lw else None)


@pytest.mark.parametrize("task,shape,n", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_finish_matches_oracle(task, shape, n, mode, oracle, dev):
    """Whole draw (paired Threefry words), with and without ancestors and pin; float32 and bfloat16 network
    output; proposal-only and weights-only calls."""
    from oracle import em
    ds, mask, emk = _mask(task, shape, dev, oracle.PRNGKey(5))
    u_off, v_off, role = _tables(oracle, mask, shape[2])
    rng = np.random.default_rng(2)
    us = rng.normal(size=(n, emk.du)).astype(np.float32)
    net = rng.normal(size=(n, emk.D)).astype(np.float32)
    v, vp = rng.normal(size=emk.dv).astype(np.float32), rng.normal(size=emk.dv).astype(np.float32)
    cx, cs, dt, sd = np.float32(0.37), np.float32(1.9), np.float32(0.002), np.float32(0.0721)
    key = oracle.PRNGKey(77)
    pin_val = rng.normal(size=emk.du).astype(np.float32)
    for A, pin_row in ((None, -1), (rng.integers(0, n, n).astype(np.int32), n // 2), (None, n - 1)):
        want_us, want_lw = em.finish(us, A, net, mode, cx, cs, dt, sd, v, vp, key, n, 0, pin_row, pin_val, u_off, v_off)
        got_us, got_lw = _finish(emk, us, A, net, mode, cx, cs, dt, sd, v, vp, key, n, 0, pin_row, pin_val, dev)
        _eq(got_us, want_us, "us_new"); _eq(got_lw, want_lw, "lw")
    # bfloat16 network output
    nb = em.to_bf16_bits(net)
    net_b = torch.from_numpy(nb.view(np.int16)).to(dev).view(torch.bfloat16)
    want_us, want_lw = em.finish(us, None, em.from_bf16_bits(nb), mode, cx, cs, dt, sd, v, vp, key, n, 0, -1, None,
                                 u_off, v_off)
    got_us, got_lw = _finish(emk, us, None, net_b, mode, cx, cs, dt, sd, v, vp, key, n, 0, -1, None, dev)
    _eq(got_us, want_us, "us_new (bf16 net)"); _eq(got_lw, want_lw, "lw (bf16 net)")
    # halves of the call
    g1, _ = _finish(emk, us, None, net, mode, cx, cs, dt, sd, None, None, key, n, 0, -1, None, dev, want_lw=False)
    _, g2 = _finish(emk, None, None, net, mode, cx, cs, dt, sd, v, vp, key, n, 0, -1, None, dev, want_us=False)
    w1, w2 = em.finish(us, None, net, mode, cx, cs, dt, sd, v, vp, key, n, 0, -1, None, u_off, v_off)
    _eq(g1, w1, "proposal only"); _eq(g2, w2, "weights only")
    # network rows reached through net_A (pmcmc_filter_step: proposal from the resampled rows of ONE evaluation)
    nA = rng.integers(0, n, n).astype(np.int32)
    g3, _ = _finish(emk, us, nA, net, mode, cx, cs, dt, sd, None, None, key, n, 0, -1, None, dev, want_lw=False, net_A=nA)
    w3, _ = em.finish(us, nA, net[nA], mode, cx, cs, dt, sd, v, vp, key, n, 0, -1, None, u_off, v_off, want_lw=False)
    _eq(g3, w3, "net_A")


@pytest.mark.parametrize("task,shape,n_total,row0,n", [("inpaint-15", (28, 28, 1), 50, 13, 21),
                                                        ("inpaint-8", (16, 16, 3), 96, 48, 48),
                                                        ("inpaint-8", (16, 16, 3), 97, 0, 40),
                                                        ("supr-4", (28, 28, 1), 33, 32, 1)])
def test_finish_row_slices_equal_the_whole_draw(task, shape, n_total, row0, n, oracle, dev):
    """A rank of a sharded ensemble: rows [row0, row0+n) of the n_total-row draw."""
    from oracle import em
    ds, mask, emk = _mask(task, shape, dev, oracle.PRNGKey(6))
    u_off, v_off, role = _tables(oracle, mask, shape[2])
    rng = np.random.default_rng(3)
    us = rng.normal(size=(n, emk.du)).astype(np.float32)
    net = rng.normal(size=(n, emk.D)).astype(np.float32)
    v, vp = rng.normal(size=emk.dv).astype(np.float32), rng.normal(size=emk.dv).astype(np.float32)
    cx, cs, dt, sd = np.float32(-0.2), np.float32(0.8), np.float32(0.01), np.float32(0.11)
    key = oracle.PRNGKey(78)
    pin_val = rng.normal(size=emk.du).astype(np.float32)
    want_us, want_lw = em.finish(us, None, net, 0, cx, cs, dt, sd, v, vp, key, n_total, row0, 0, pin_val, u_off, v_off)
    got_us, got_lw = _finish(emk, us, None, net, 0, cx, cs, dt, sd, v, vp, key, n_total, row0, 0, pin_val, dev)
    _eq(got_us, want_us, "us_new"); _eq(got_lw, want_lw, "lw")


@pytest.mark.parametrize("task,shape,n", CASES[:3])
def test_transition_logpdf_matches_oracle(task, shape, n, oracle, dev):
    from fbs_amd import _lib
    from oracle import em
    ds, mask, emk = _mask(task, shape, dev, oracle.PRNGKey(7))
    u_off, v_off, role = _tables(oracle, mask, shape[2])
    rng = np.random.default_rng(4)
    us = rng.normal(size=(n, emk.du)).astype(np.float32)
    net = rng.normal(size=(n, emk.D)).astype(np.float32)
    u = rng.normal(size=emk.du).astype(np.float32)
    for mode in (0, 1):
        want = em.transition_logpdf(us, net, mode, 0.3, 1.2, 0.004, 0.09, u, u_off)
        t = lambda a: torch.from_numpy(a).to(dev)
        ust, nt, ut = t(us), t(net), t(u)
        lw = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.call("fbsmi_em_transition_logpdf", emk.ref, ust.data_ptr(), nt.data_ptr(), 0, mode, 0.3, 1.2, 0.004, 0.09,
                  ut.data_ptr(), n, lw.data_ptr(), 0)
        torch.cuda.synchronize()
        _eq(_np(lw), want, f"transition_logpdf mode {mode}")


def test_config5_shard_shape_full_size(oracle, dev):
    """CelebA-64 inpaint-32 (du = 3072, dv = 9216 floats), the N = 2048 rows one GPU owns of BASELINE config 5's
    16 384: the whole draw (paired words) and as rank 3's slice of the 16 384-row draw, bit for bit."""
    from oracle import em
    shape, n = (64, 64, 3), 2048
    ds, mask, emk = _mask("inpaint-32", shape, dev, oracle.PRNGKey(8))
    assert (emk.du, emk.dv) == (3072, 9216)
    u_off, v_off, role = _tables(oracle, mask, 3)
    rng = np.random.default_rng(5)
    us = rng.normal(size=(n, emk.du)).astype(np.float32)
    net = rng.normal(size=(n, emk.D)).astype(np.float32)
    v, vp = rng.normal(size=emk.dv).astype(np.float32), rng.normal(size=emk.dv).astype(np.float32)
    A = rng.integers(0, n, n).astype(np.int32)
    cx, cs, dt, sd = np.float32(1.3), np.float32(2.6), np.float32(0.002), np.float32(0.0721)
    key = oracle.PRNGKey(79)
    pin_val = rng.normal(size=emk.du).astype(np.float32)
    want_us, want_lw = em.finish(us, A, net, 0, cx, cs, dt, sd, v, vp, key, n, 0, 5, pin_val, u_off, v_off)
    got_us, got_lw = _finish(emk, us, A, net, 0, cx, cs, dt, sd, v, vp, key, n, 0, 5, pin_val, dev)
    _eq(got_us, want_us, "us_new"); _eq(got_lw, want_lw, "lw")
    want_us, _ = em.finish(us, A, net, 0, cx, cs, dt, sd, v, vp, key, 16384, 3 * n, -1, None, u_off, v_off, want_lw=False)
    got_us, _ = _finish(emk, us, A, net, 0, cx, cs, dt, sd, v, vp, key, 16384, 3 * n, -1, None, dev, want_lw=False)
    _eq(got_us, want_us, "us_new (rank 3 of 8)")
    want = em.concat(us, A, vp, role)
    from fbs_amd import _lib
    t = lambda a: torch.from_numpy(a).to(dev)
    ust, At, vpt = t(us), t(A), t(vp)
    img = torch.empty((n, emk.D), dtype=torch.float32, device=dev)
    _lib.call("fbsmi_em_concat", emk.ref, ust.data_ptr(), At.data_ptr(), vpt.data_ptr(), n, 0, img.data_ptr(), 0)
    _eq(_np(img), want, "concat")


# ------------------------------------------------------------------------------------------------------------------
# the closure tier with a ScoreBridge: a whole gibbs_kernel / bootstrap_filter / pmcmc_filter_step against a numpy
# restatement driven by the same (deterministic, elementwise) stand-in network
# ------------------------------------------------------------------------------------------------------------------
def _toy_net(x, t):
    """An elementwise 'network' that numpy reproduces bit for bit (separately rounded torch ops)."""
    y = x * 0.75
    y = y + (0.1 * float(t))
    return y


def _toy_net_np(x, t):
    return ((x * np.float32(0.75)).astype(np.float32) + np.float32(0.1 * float(t))).astype(np.float32)


def _bridge(dev, task, shape, T, mode="score"):
    from fbs_amd.images import ImageRestore
    from fbs_amd.score import ScoreBridge
    from fbs_amd.sdes import StationaryLinLinearSDE
    Tend = 2.0
    ts = np.linspace(0, Tend, T + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=Tend)
    ds = ImageRestore(task, shape, device=dev)
    return ds, ScoreBridge(_toy_net, ds, sde, ts, chunk=7, mode=mode), sde, ts


def _oracle_forward(O, sb, u_off, v_off, role, key, us_star, bs, vs, us0, lw0):
    """oracle/em.py's restatement of csmc.forward_pass over the image closures, with the stand-in network."""
    from oracle import em
    return em.forward_pass(key, us_star, bs, vs, us0, lw0, sb.ts, sb._coef, lambda img, t: _toy_net_np(img, sb.T - float(t)),
                           sb.dt, u_off, v_off, role)


@pytest.mark.parametrize("task,shape", [("inpaint-15", (28, 28, 1)), ("inpaint-8", (16, 16, 3))])
def test_forward_pass_with_score_bridge_equals_oracle(task, shape, oracle, dev):
    from fbs_amd.samplers.csmc.csmc import forward_pass
    from fbs_amd.samplers.csmc.resamplings import killing
    T, n = 9, 48
    ds, sb, sde, ts = _bridge(dev, task, shape, T)
    mask = ds.gen_mask(oracle.PRNGKey(11))
    u_off, v_off, role = _tables(oracle, mask, shape[2])
    rng = np.random.default_rng(9)
    p, q, c = u_off.size // shape[2], v_off.size // shape[2], shape[2]
    us_star = rng.normal(size=(T + 1, p * c)).astype(np.float32)
    vs = rng.normal(size=(T + 1, q * c)).astype(np.float32)
    bs = rng.integers(0, n + 1, T + 1).astype(np.int32)
    us0 = rng.normal(size=(n + 1, p * c)).astype(np.float32)
    lw0 = rng.normal(size=n + 1).astype(np.float32)
    key = oracle.PRNGKey(21)
    t = lambda a: torch.from_numpy(a).to(dev)
    calls = {"n": 0}
    inner = sb.score_fn

    def counted(x, tt):
        calls["n"] += 1
        return inner(x, tt)

    sb.score_fn = counted
    As, lws, uss = forward_pass(key, t(us_star).reshape(T + 1, p, c), bs, t(vs).reshape(T + 1, q, c), ts,
                                lambda k_, m_: t(us0).reshape(n + 1, p, c), lambda v0, u0s, v1, **kw: t(lw0),
                                sb.transition_sampler, sb.likelihood_logpdf, killing, n, mask_=mask)
    assert calls["n"] == T * -(-(n + 1) // 7)          # one network evaluation per step (chunks of at most 7)
    wAs, wlw, wus = _oracle_forward(oracle, sb, u_off, v_off, role, key, us_star, bs, vs, us0, lw0)
    _eq(_np(As), wAs, "As")
    _eq(_np(uss[-1]).reshape(n + 1, -1), wus, "final particles")
    _eq(_np(lws[-1]), wlw, "final log-weights")
    # the closures called one by one (the reference's protocol) give the same step as the fused kernel
    k = 3
    usk, A = uss[k], As[k]
    from fbs_amd import ops
    up = ops.take_rows(usk, A)
    kt = oracle.split(oracle.split(oracle.split(key, 2)[1], T)[k], 2)[1]
    a = sb.transition_sampler(up, t(vs[k]).reshape(q, c), ts[k], kt, mask_=mask)
    b = sb.likelihood_logpdf(t(vs[k + 1]).reshape(q, c), up, t(vs[k]).reshape(q, c), ts[k], mask_=mask)
    a = ops.set_row(a, int(bs[k + 1]), t(us_star[k + 1]).reshape(p, c))
    _eq(_np(a), _np(uss[k + 1]), "closure transition_sampler vs fused step")
    _eq(_np(ops.normalise(b, log_space=True)), _np(lws[k + 1]), "closure likelihood_logpdf vs fused step")


def test_gibbs_kernel_score_and_drift_bridges_run_fused(oracle, dev):
    """gibbs_kernel (eb x ef) over the image closures: score model (inpainting.py) and SB drift model
    (sb_imgs/supr.py: backward net as drift, forward path by euler_maruyama with the forward net, supr-4,
    explicit_final=True)."""
    from fbs_amd import ops
    from fbs_amd.samplers import gibbs_kernel
    T, n = 6, 32
    for task, shape, mode in (("inpaint-15", (28, 28, 1), "score"), ("supr-4", (28, 28, 1), "drift")):
        ds, sb, sde, ts = _bridge(dev, task, shape, T, mode)
        ds.sr_random = False                                                     # sb_imgs/supr.py:56
        sb.fwd_drift_fn = lambda x, tt: -0.5 * x
        mask = ds.gen_mask(oracle.PRNGKey(12))
        img = ops.uniform(oracle.PRNGKey(13), shape, device=dev)
        _, y0 = ds.unpack(img, mask)
        x0 = torch.zeros(ds.unobs_shape, device=dev)
        bs = np.zeros(T + 1, np.int32)
        for eb, ef in ((True, True), (True, False)):
            out = gibbs_kernel(oracle.PRNGKey(14), x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n,
                               sb.transition_sampler, sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False,
                               explicit_backward=eb, explicit_final=ef, mask_=mask)
            x0n, usn, bsn, acc = out
            assert x0n.shape == tuple(ds.unobs_shape) and usn.shape == (T + 1,) + tuple(ds.unobs_shape)
            assert torch.isfinite(usn).all() and bsn.shape == (T + 1,)
            again = gibbs_kernel(oracle.PRNGKey(14), x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n,
                                 sb.transition_sampler, sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False,
                                 explicit_backward=eb, explicit_final=ef, mask_=mask)
            assert all(torch.equal(a, b) for a, b in zip(out, again)), "the sweep is not reproducible"


def test_filters_with_score_bridge_equal_the_closure_by_closure_loop(oracle, dev):
    """bootstrap_filter and pmcmc_filter_step take the fused step when handed a ScoreBridge's closures; handed the same
    closures wrapped in plain functions (so that the bridge is not recognised) they run the reference's closure-by-
    closure loop.  Both must agree bit for bit."""
    from fbs_amd import ops
    from fbs_amd.samplers import stratified
    from fbs_amd.samplers.smc import bootstrap_filter, pmcmc_filter_step
    T, n = 7, 40
    ds, sb, sde, ts = _bridge(dev, "inpaint-8", (16, 16, 3), T)
    mask = ds.gen_mask(oracle.PRNGKey(15))
    rng = np.random.default_rng(10)
    vs = torch.from_numpy(rng.normal(size=(T + 1,) + (16 * 16 - 64, 3)).astype(np.float32)).to(dev)
    init = lambda k_, v0, m_: ops.normal(k_, (m_,) + tuple(ds.unobs_shape), device=dev)
    tsamp = lambda *a, **kw: sb.transition_sampler(*a, **kw)
    like = lambda *a, **kw: sb.likelihood_logpdf(*a, **kw)
    key = oracle.PRNGKey(16)
    for return_last in (True, False):
        a = bootstrap_filter(sb.transition_sampler, sb.likelihood_logpdf, vs, ts, init, key, n, stratified,
                             return_last=return_last, mask_=mask)
        b = bootstrap_filter(tsamp, like, vs, ts, init, key, n, stratified, return_last=return_last, mask_=mask)
        _eq(_np(a[0]), _np(b[0]), f"bootstrap_filter samples (return_last={return_last})")
        _eq(_np(a[1]).reshape(1), _np(b[1]).reshape(1), "bootstrap_filter nell")
    u0s = init(oracle.PRNGKey(17), None, n)
    a = pmcmc_filter_step(key, vs, u0s, ts, sb.transition_sampler, sb.likelihood_logpdf, stratified, n, mask_=mask)
    b = pmcmc_filter_step(key, vs, u0s, ts, tsamp, like, stratified, n, mask_=mask)
    _eq(_np(a[0]), _np(b[0]), "pmcmc_filter_step particles")
    _eq(_np(a[1]).reshape(1), _np(b[1]).reshape(1), "pmcmc_filter_step log_ell")


def test_closure_cache_is_not_fooled_by_recycled_storage(oracle, dev):
    """the shared-network cache must key on the tensors themselves, not on addresses."""
    from fbs_amd import ops
    ds, sb, sde, ts = _bridge(dev, "inpaint-8", (16, 16, 3), 4)
    sb.score_fn = lambda x, t: x * 0.75 + x.mean(dim=(1, 2, 3), keepdim=True)   # couples the hidden pixels into every output
    mask = ds.gen_mask(oracle.PRNGKey(18))
    n = 16
    vp = ops.normal(oracle.PRNGKey(19), (16 * 16 - 64, 3), device=dev)
    outs = []
    for s in range(3):
        us = ops.normal(oracle.PRNGKey(100 + s), (n,) + tuple(ds.unobs_shape), device=dev)   # likely the same address
        outs.append(sb.likelihood_logpdf(vp, us, vp, ts[1], mask_=mask).clone())
        del us
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    us = ops.normal(oracle.PRNGKey(100), (n,) + tuple(ds.unobs_shape), device=dev)
    a = sb.likelihood_logpdf(vp, us, vp, ts[1], mask_=mask).clone()
    us.add_(1.0)                                                                             # in-place: new version
    b = sb.likelihood_logpdf(vp, us, vp, ts[1], mask_=mask)
    assert not torch.equal(a, b)
