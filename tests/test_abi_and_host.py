"""CPU: the C-ABI libraries load and export every symbol include/fbsmi.h, fbsmi_nn.h and fbsmi_dist.h declare; host-side logic
(key splitting, SDE coefficients, table builder) agrees with the oracle / closed forms; the product
refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import toy_2d, toy_4d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "fbsmi.h")).read() + open(os.path.join(ROOT, "include", "fbsmi_nn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fbsmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fbs_amd import _lib
    L = ctypes.CDLL(_lib.build())
    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fbsmi.h but not exported"
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    assert _lib.lib().fbsmi_abi_version() == 1


def test_dist_library_exports_every_declared_symbol():
    """include/fbsmi_dist.h (the multi-GPU exchange steps): the library links librccl and libfbsmi, loads without a GPU, and
    refuses impossible shardings at creation (no device call before the checks)."""
    from fbs_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fbsmi_dist.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(fbsmi_dist_[a-z0-9_]+)\s*\(", text)))
    L = ctypes.CDLL(_lib.build_dist())
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fbsmi_dist.h but not exported"
    assert set(_lib.DIST_SIGNATURES) == set(names), set(_lib.DIST_SIGNATURES) ^ set(names)
    D = _lib.dist_lib()
    assert D.fbsmi_dist_abi_version() == 1
    h = ctypes.c_void_p()
    for rank, world, rows in ((0, 8, 9), (2, 2, 10), (0, 17, 100), (0, 4, 3)):   # 9 rows over 8 ranks: the last would own none
        assert D.fbsmi_dist_create(None, rank, world, rows, ctypes.byref(h)) == -1 and not h.value
        assert len(D.fbsmi_dist_last_error()) > 0


def test_host_key_split_matches_oracle_and_golden(oracle):
    from fbs_amd import ops
    g = np.load(os.path.join(ROOT, "tests", "golden", "keys_slice.npz"))
    mine = ops.split(ops.PRNGKey(666), 1000)
    np.testing.assert_array_equal(mine[g["rows"]], g["keys"])
    for seed, num in ((0, 1), (1, 2), (2**40 + 5, 7), (666, 33)):
        np.testing.assert_array_equal(ops.split(ops.PRNGKey(seed), num), oracle.split(oracle.PRNGKey(seed), num))


def test_no_gpu_means_error_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fbs_amd import ops
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    with pytest.raises(RuntimeError):
        ops.cumsum(torch.ones(8))
    with pytest.raises(RuntimeError):
        ops.uniform(ops.PRNGKey(0), (4,))
    toy = toy_2d()
    with pytest.raises(RuntimeError):
        fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(-0.5, 1.0),
                                     np.linspace(0, 1, 11), 1)


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = "import sys; import fbs_amd, fbs_amd.samplers.csmc.csmc; print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stdout + out.stderr
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fbs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


# ---- fbs/sdes/linear.py restated (tests/test_sdes.py:18-90) -------------------------------------
def test_sde_discretisation_closed_forms():
    from fbs_amd.sdes import make_linear_sde, make_ou_sde, StationaryConstLinearSDE, StationaryLinLinearSDE
    a, b = -0.5, 1.0
    disc, score, _ = make_linear_sde(StationaryConstLinearSDE(a, b))
    F, Q = disc(50.0, 0.0)
    np.testing.assert_allclose([F, Q], [0.0, b ** 2 / (-2 * a)], atol=1e-8)           # stationarity
    # lin SDE with a constant beta == const SDE (test_sdes.py:64-77)
    beta = 1.3
    lin = StationaryLinLinearSDE(beta_min=beta, beta_max=beta, t0=0.0, T=2.0)
    const = StationaryConstLinearSDE(a=-0.5 * beta, b=np.sqrt(beta))
    for t, s in ((0.7, 0.1), (2.0, 0.0)):
        np.testing.assert_allclose(make_linear_sde(lin)[0](t, s), make_linear_sde(const)[0](t, s), rtol=1e-12)
    # closed-form alpha of the linear schedule (test_sdes.py:79-90)
    lin = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=2.0)
    t = 1.3
    bint = 0.02 * t + 0.5 * (5.0 - 0.02) / 2.0 * t ** 2
    np.testing.assert_allclose(lin.beta_integral(t, 0.0), bint, rtol=1e-12)
    np.testing.assert_allclose(lin.mean(t, 0.0, 2.0), 2.0 * np.exp(-0.5 * bint), rtol=1e-12)
    np.testing.assert_allclose(lin.variance(t, 0.0), 1 - np.exp(-bint), rtol=1e-12)
    # make_ou_sde == make_linear_sde(const) exactly (test_sdes.py:135-163)
    d_ou, s_ou, _ = make_ou_sde(a, b)
    assert d_ou(0.37) == disc(0.37, 0.0)
    x = torch.tensor([0.3, -1.2])
    x0 = torch.tensor([1.0, 2.0])
    assert torch.equal(s_ou(x, 0.37, x0), score(x, 0.37, x0, 0.0))


def test_bridge_drift_is_the_gradient_of_log_h():
    """bridge_drift = drift + b^2 d/dx log N(target; F x, Q) (linear.py:36-45) by finite differences."""
    from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE
    from fbs_amd.sdes.linear import discretise_linear_sde_np
    for sde in (StationaryConstLinearSDE(-0.5, 1.0), StationaryLinLinearSDE(0.02, 5.0, 0.0, 2.0)):
        T, t, x, target = 2.0, 0.6, 0.4, 1.7

        def log_h(xx):
            F, Q = discretise_linear_sde_np(sde, T, t)
            return -0.5 * (target - F * xx) ** 2 / Q

        g = (log_h(x + 1e-6) - log_h(x - 1e-6)) / 2e-6
        want = sde.drift(x, t) + sde.dispersion(t) ** 2 * g
        np.testing.assert_allclose(sde.bridge_drift(x, t, target, T), want, rtol=1e-6)


@pytest.mark.parametrize("toy", [toy_2d, toy_4d])
def test_table_builder_matches_reference_closures(toy, oracle):
    """G z + g reproduces reverse_drift of experiments/toy/gp_gibbs.py:73-95 evaluated directly
    (Cholesky solve of the marginal covariance) in float64, and equals the oracle's own derivation."""
    from fbs_amd.linear_gaussian import lg_tables
    from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE
    toy = toy()
    ts = np.linspace(0, 2, 41)
    for sde, osde in ((StationaryConstLinearSDE(-0.5, 1.0), oracle.sde_const(-0.5, 1.0)),
                      (StationaryLinLinearSDE(0.02, 4.0, 0.0, 2.0), oracle.sde_lin(0.02, 4.0, 0.0, 2.0))):
        tab = lg_tables(toy["m0"], toy["cov0"], sde, ts, toy["du"])
        otab = oracle.lg_tables_f64(toy["m0"], toy["cov0"], osde, ts, toy["du"])
        for k in ("G", "g", "sd", "lognorm", "F", "sqQ"):
            np.testing.assert_allclose(tab[k], otab[k], rtol=1e-12, atol=1e-14)
        rng = np.random.default_rng(0)
        D = toy["m0"].size
        for k in (0, 13, 39):
            z = rng.normal(size=D)
            t_fwd = ts[-1] - ts[k]
            Ft, Qt = osde["FQ"](t_fwd, ts[0])
            cov = Ft ** 2 * toy["cov0"] + Qt * np.eye(D)
            score = -np.linalg.solve(cov, z - Ft * toy["m0"])
            want = -osde["a"](t_fwd) * z + osde["b"](t_fwd) ** 2 * score
            np.testing.assert_allclose(tab["G"][k] @ z + tab["g"][k], want, rtol=1e-10)


def test_nn_kernels_refuse_unsupported_shapes_loudly():
    """include/fbsmi_nn.h: shape checks run on the host before any launch (no GPU needed to see them)."""
    from fbs_amd import _lib
    p = 4096   # never dereferenced: every call below fails its argument check first
    with pytest.raises(NotImplementedError):
        _lib.call("fbsmi_nn_linear_attention", p, p, 0, 1, 16, 4, 16, None)          # dim_head != 32
    with pytest.raises(NotImplementedError):
        _lib.call("fbsmi_nn_groupnorm_silu", p, p, 0, 1, 16, 12, 8, p, p, 1e-6, None, None, None, None, None, None)   # C % (8 * groups)
    with pytest.raises(NotImplementedError):
        _lib.call("fbsmi_nn_channel_layernorm", p, p, 1, 10, 24, p, 1e-5, None, None, None)      # C / 8 = 3 is not a power of two
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_nn_linear_attention", None, p, 0, 1, 16, 4, 32, None)       # null input
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_nn_groupnorm_silu", p, p, 2, 1, 16, 64, 8, p, p, 1e-6, None, None, None, None, None, None)   # unknown dtype
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_nn_bias_add", p, 1, 10, 12, p, None)                         # C not a multiple of 8
    with pytest.raises(RuntimeError):
        _lib.call("fbsmi_nn_pixel_shuffle", p, p, 1, 2, 7, 7, 4, 2, None, None)       # c not a multiple of 8


def test_bridge_of_recognises_only_the_bridges_own_pair():
    """the fused score-network step replaces (transition_sampler, likelihood_logpdf); another method of the same
    bridge, a swapped pair or a subclass override must not be mistaken for them."""
    from fbs_amd.score import ScoreBridge, bridge_of
    sb = object.__new__(ScoreBridge)
    assert bridge_of(sb.transition_sampler, sb.likelihood_logpdf) is sb
    assert bridge_of(sb.transition_sampler, sb.likelihood_logpdf, sb.transition_logpdf) is sb
    assert bridge_of(sb.transition_sampler, sb.transition_logpdf) is None          # the wrong weight function
    assert bridge_of(sb.likelihood_logpdf, sb.transition_sampler) is None          # swapped
    assert bridge_of(sb.transition_sampler, object.__new__(ScoreBridge).likelihood_logpdf) is None   # two bridges

    class Mine(ScoreBridge):
        def likelihood_logpdf(self, *a, **k):
            return None
    m = object.__new__(Mine)
    assert bridge_of(m.transition_sampler, m.likelihood_logpdf) is None            # an override is not the built-in
    assert bridge_of(lambda *a: None, lambda *a: None) is None


def test_same_sde_compares_coefficients_of_any_kind():
    from fbs_amd.samplers.gibbs import _same_sde
    from fbs_amd.sdes import StationaryConstLinearSDE, StationaryLinLinearSDE
    a, b = StationaryConstLinearSDE(a=-0.5, b=1.0), StationaryConstLinearSDE(a=-0.5, b=1.0)
    assert _same_sde(a, a) and _same_sde(a, b) and not _same_sde(a, StationaryConstLinearSDE(a=-0.4, b=1.0))
    assert not _same_sde(a, StationaryLinLinearSDE(0.02, 5.0, 0.0, 2.0))
    a.extra, b.extra = np.arange(3.0), np.arange(3.0)          # array attributes no longer raise
    assert _same_sde(a, b)
    b.extra = np.arange(4.0)
    assert not _same_sde(a, b)
