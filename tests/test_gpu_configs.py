"""GPU: BASELINE configs 3, 4 and 5 AT SIZE through gibbs_kernel on the fused score-network step
(fbs_amd/image_configs.py: synthetic image, randomly initialised UNet dim 64 in bf16 -- no checkpoints exist here).

  config 3: MNIST inpaint-15, N = 4096 (+1: explicit_final), T = 1000                      -- the whole configuration
  config 4: MNIST Schrodinger-bridge supr-4, the 2048 (+1) particles one of 4 GPUs owns, T = 50
  config 5: CelebA-64 inpaint-32, the 2048 (+1) particles one of 8 GPUs owns, T = 1000 (the whole sweep of that GPU's share:
            80-95 s of network time in bf16)

The network has no reference here (parity unpinned), so what is checked at these sizes is what
does not depend on it: the LAST SMC step of the sweep is captured with the network output it actually saw and replayed
through the numpy oracle (oracle/em.py) -- network input, proposal, pin and log-weights bit for bit --, the weights are
normalised, the reference trajectory and the acceptance flags have the reference's shapes, and the network ran once
per step and chunk."""
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


def _run(name, n, oracle, dev, nsteps=None, dtype="bf16"):
    from fbs_amd import image_configs, ops
    from oracle import em
    c = image_configs.make(name, dev, dtype=dtype, nsteps=nsteps)
    T = c.cfg["nsteps"]
    c.sb.capture = {}
    rng = np.random.default_rng(1)
    bs = rng.integers(0, n, T + 1).astype(np.int32)
    x0n, us_star, bs_next, acc = image_configs.gibbs_sweep(c, ops.PRNGKey(2024), n, bs_star=bs)
    torch.cuda.synchronize()
    rows = n + 1                                                         # explicit_final (gibbs.py:133-134)
    chunks = -(-rows // c.cfg["chunk"])
    fwd_calls = 2 * T if c.cfg["mode"] == "drift" else 0                 # two forward paths by euler_maruyama (supr.py:137)
    assert c.timers["calls"] == (T + 1) * chunks + fwd_calls, (c.timers["calls"], T, chunks)
    p, ch = c.ds.unobs_shape
    assert x0n.shape == (p, ch) and us_star.shape == (T + 1, p, ch) and bs_next.shape == (T + 1,) and acc.shape == (T + 1,)
    assert torch.isfinite(us_star).all() and acc.dtype == torch.bool
    assert int(bs_next.min()) >= 0 and int(bs_next.max()) < n
    # replay the last SMC step through the oracle, with the network output the kernels saw
    cap = c.sb.capture
    emk = cap["em"]
    u_off, v_off, role = em.element_tables(_np(c.mask.unobs_inds_ravelled), _np(c.mask.obs_inds_ravelled), ch)
    us, A = _np(cap["us"]), _np(cap["A"])
    assert us.shape == (rows, emk.du) and A.shape == (rows,) and A.min() >= 0 and A.max() < rows
    vp, v = _np(cap["v_prev"]).reshape(-1), _np(cap["v"]).reshape(-1)
    if dtype == "bf16":
        img_bits = _np(cap["img"].reshape(rows, -1).view(torch.int16)).view(np.uint16)
        _eq(img_bits, em.to_bf16_bits(em.concat(us, A, vp, role)), "network input of the last step")
        net = em.from_bf16_bits(_np(cap["net"].view(torch.int16)).view(np.uint16))
    else:   # the reference's precision: float32 network input and output (fbs/nn/unet.py:85-86)
        assert cap["img"].dtype == torch.float32 and cap["net"].dtype == torch.float32
        _eq(_np(cap["img"].reshape(rows, -1)), em.concat(us, A, vp, role), "network input of the last step")
        net = _np(cap["net"].reshape(rows, -1))
    mode, cx, cs, sd = cap["coef"]
    pin_row, pin_val = cap["pin"]
    assert pin_row == int(bs[T])
    want_us, want_lw = em.finish(us, A, net, mode, np.float32(cx), np.float32(cs), np.float32(c.sb.dt), np.float32(sd), v,
                                 vp, cap["key"], rows, 0, pin_row, _np(pin_val).reshape(-1), u_off, v_off)
    _eq(_np(cap["us_new"]), want_us, "particles of the last step")
    _eq(_np(cap["lw"]), want_lw, "log-weights of the last step")
    _eq(_np(cap["us_new"])[pin_row], _np(pin_val).reshape(-1), "pinned reference particle")
    assert np.isfinite(want_lw).all()
    return c


def test_config3_whole_configuration(oracle, dev):
    _run("c3", 4096, oracle, dev)


def test_config4_one_gpu_share(oracle, dev):
    c = _run("c4", 2048, oracle, dev)
    assert c.cfg["mode"] == "drift" and c.cfg["nsteps"] == 50 and c.sb.capture["coef"][0] == 1


def test_config5_one_gpu_share(oracle, dev):
    _run("c5", 2048, oracle, dev)          # the configuration's own T = 1000 (about 95 s of network time in bf16)


@pytest.mark.parametrize("name,n,nsteps", [("c3", 4096, 40), ("c5", 2048, 12)])
def test_configs_in_float32_the_references_precision(name, n, nsteps, oracle, dev):
    """The same sweeps with the network in float32, as the reference computes it (bf16 autocast is this build's fast path):
    per-step shapes as in the configuration, a reduced number of steps (a float32 step costs about four bf16 ones)."""
    _run(name, n, oracle, dev, nsteps=nsteps, dtype="f32")
