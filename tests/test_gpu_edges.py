"""GPU: edge cases through the C ABI -- empty inputs, single elements, argument errors, capacity
limits (the API must fail loudly, SURVEY section 7 hard part 5)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import toy_2d

pytestmark = pytest.mark.gpu


def test_empty_and_single_element_inputs(oracle, dev):
    from fbs_amd import ops
    key = oracle.PRNGKey(0)
    assert ops.uniform(key, (0,), device=dev).numel() == 0
    assert ops.normal(key, (0, 3), device=dev).shape == (0, 3)
    assert ops.random_bits(key, (0,), device=dev).numel() == 0
    assert ops.cumsum(torch.empty(0, device=dev)).numel() == 0
    one = torch.tensor([0.25], device=dev)
    assert ops.cumsum(one).item() == 0.25 and ops.tree_sum(one).item() == 0.25
    assert ops.logsumexp(torch.tensor([-3.0], device=dev)).item() == -3.0
    assert ops.take_rows(torch.zeros((4, 2), device=dev), torch.empty(0, dtype=torch.int32, device=dev)).shape == (0, 2)
    q = torch.empty(0, device=dev)
    assert ops.searchsorted(torch.tensor([0.5, 1.0], device=dev), q).numel() == 0
    # scalar draws follow the size-1 counter layout
    assert ops.uniform(key, (), device=dev).item() == float(oracle.uniform(key, ()))
    # -inf log-weights (zero-probability particles) normalise to exact zeros
    lw = torch.tensor([0.0, -float("inf"), -1.0], device=dev)
    w = ops.normalise(lw).cpu().numpy()
    np.testing.assert_array_equal(w.view(np.uint32), oracle.normalise(lw.cpu().numpy(), False).view(np.uint32))
    assert w[1] == 0.0


def test_argument_errors_are_loud(oracle, dev):
    from fbs_amd import _lib, ops
    from fbs_amd.samplers.csmc import resamplings as CR
    from fbs_amd.samplers.gibbs import force_move
    w = torch.full((8,), 0.125, device=dev)
    with pytest.raises(RuntimeError):
        CR.killing(oracle.PRNGKey(0), w, 8, 0, True)          # i out of range
    with pytest.raises(RuntimeError):
        CR.multinomial(oracle.PRNGKey(0), w, 0, -1, True)     # j out of range
    with pytest.raises(RuntimeError):
        force_move(oracle.PRNGKey(0), w, 9)
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_cumsum", None, 5, None, None, None)
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_math_map", 99, w.data_ptr(), None, 8, w.data_ptr(), None)
    assert b"bad arguments" in _lib.lib().fbsmi_last_error()
    with pytest.raises(RuntimeError):
        ops.cumsum(torch.ones(4))                                # host tensor: no CPU path


def test_capacity_limits(oracle, dev):
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    toy = toy_2d()
    br = fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(-0.5, 1.0),
                                      np.linspace(0, 1, 1001), 1, device=dev)
    with pytest.raises(NotImplementedError):
        br.sweep_handle(5_000_000)                               # > 4M particles per device
    long = fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(-0.5, 1.0),
                                        np.linspace(0, 1, 6001), 1, device=dev)
    with pytest.raises(NotImplementedError):
        long.sweep_handle(4_000_000, explicit_backward=False)    # (T+1, N, du) path storage > device memory
    with pytest.raises(RuntimeError):
        br.sweep_handle(0)
    # du, dv > 128 exceed the matrix-core drift kernel: the fused engine refuses, the dispatcher falls back to the
    # closure tier
    rng = np.random.default_rng(0)
    A = rng.normal(size=(260, 260))
    big = fbs_amd.LinearGaussianBridge(np.zeros(260), A @ A.T / 260 + np.eye(260), StationaryConstLinearSDE(-0.5, 1.0),
                                       np.linspace(0, 1, 4), 130, device=dev)
    with pytest.raises(NotImplementedError):
        big.sweep_handle(64)
    from fbs_amd.samplers import gibbs_kernel
    out = gibbs_kernel(oracle.PRNGKey(1), torch.zeros(130, device=dev), torch.zeros(130, device=dev), None,
                       np.zeros(4, np.int32), np.linspace(0, 1, 4), big.fwd_sampler, big.sde, big.unpack, 32,
                       big.transition_sampler, big.transition_logpdf, big.likelihood_logpdf)
    assert out[1].shape == (4, 130) and torch.isfinite(out[1]).all()


def test_large_dimension_closures_match_oracle(oracle, dev):
    """du = dv = 20 (the d-dimensional GP toy of gp_gibbs.py:32-58 has joint dimension 2d): the fused engine
    (through gibbs_kernel's dispatch) and the closure tier (plain lambdas hide the bridge), both bit-exact
    against the oracle."""
    import fbs_amd
    from fbs_amd.samplers import gibbs_kernel
    from fbs_amd.sdes import StationaryConstLinearSDE
    from helpers import oracle_model_from
    d = 20
    zs = np.linspace(0, 5, d)
    cov = np.exp(-np.abs(zs[None, :] - zs[:, None]))                       # gp_gibbs.py:39-40
    joint_cov = np.block([[cov, cov], [cov, cov + np.eye(d)]])             # :55-57
    ts = np.linspace(0, 1, 9)
    br = fbs_amd.LinearGaussianBridge(np.zeros(2 * d), joint_cov, StationaryConstLinearSDE(-0.5, 1.0), ts, d, device=dev)
    om = oracle_model_from(oracle, br)
    rng = np.random.default_rng(3)
    y0 = rng.normal(size=d).astype(np.float32)
    x0 = rng.normal(size=d).astype(np.float32)
    bs = rng.integers(0, 48, 9).astype(np.int32)
    key = oracle.PRNGKey(17)
    for ef in (False, True):
        assert br.fused_sweep_supported(48, ef)
        want = oracle.gibbs_kernel_lg(om, key, x0, y0, bs, 48, True, ef)
        for tier in ("fused", "closure"):
            ls = br.likelihood_logpdf if tier == "fused" else (lambda *a, **k: br.likelihood_logpdf(*a, **k))
            got = gibbs_kernel(key, torch.from_numpy(x0).to(dev), torch.from_numpy(y0).to(dev), None, bs, ts, br.fwd_sampler,
                               br.sde, br.unpack, 48, br.transition_sampler, br.transition_logpdf, ls, explicit_final=ef)
            for a, b in zip(got, want):
                a = a.cpu().numpy()
                assert np.array_equal(a.view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), f"{tier}, explicit_final={ef}"
