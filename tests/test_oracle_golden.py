"""CPU: pin the oracle.  Bit-level against the reference's only bit-level fixture
(experiments/keys.npy slice) and the Threefry known-answer vectors; numerically against float64
numpy / scipy; distributionally against the closed-form targets of the reference's own tests."""
import os

import numpy as np
import pytest
import scipy.special as sp

from helpers import toy_2d

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_split_reproduces_reference_keys(oracle):
    g = np.load(os.path.join(GOLD, "keys_slice.npz"))
    mine = oracle.split(oracle.PRNGKey(666), 1000)
    np.testing.assert_array_equal(mine[g["rows"]], g["keys"])
    assert mine.dtype == np.uint32 and mine.shape == (1000, 2)


def test_threefry_known_answers(oracle):
    # Random123 threefry2x32 (20 rounds) known-answer vectors
    kat = [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6b200159, 0x99ba4efe)),
           ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
           ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]
    for key, ctr, out in kat:
        got = oracle.threefry2x32(np.array(key, np.uint32), ctr[0], ctr[1])
        assert tuple(int(x) for x in got) == out


def test_random_bits_layout(oracle):
    key = oracle.PRNGKey(7)
    for n in (1, 2, 5, 8):
        bits = oracle.random_bits(key, n)
        half = (n + 1) // 2
        for i in range(half):
            j = i + half
            o = oracle.threefry2x32(key, i, j if j < n else 0)
            assert bits[i] == o[0]
            if j < n:
                assert bits[j] == o[1]


def test_math_spec_accuracy(oracle):
    x = np.linspace(-87, 88, 400001).astype(np.float32)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(oracle.exp(x) - ref) / ref) < 7e-8           # correctly rounded or 1 ulp
    x = np.exp(np.linspace(-80, 80, 400001)).astype(np.float32)
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(oracle.log(x) - ref) / np.maximum(np.abs(ref), 1e-30)) < 2e-7
    x = np.linspace(-0.999999, 0.999999, 400001).astype(np.float32)
    ref = sp.erfinv(x.astype(np.float64))
    assert np.nanmax(np.abs(oracle.erfinv(x) - ref) / np.maximum(np.abs(ref), 1e-30)) < 1e-5
    assert oracle.exp(np.array([-np.inf, -88.0], np.float32)).tolist() == [0.0, 0.0]
    assert oracle.log(np.array([0.0], np.float32))[0] == -np.inf


def test_exp_is_monotone_over_all_of_float32(oracle):
    """The identity max_i exp(x_i) == exp(max_i x_i) that the killing kernel relies on."""
    assert oracle.lib().orc_exp_monotone_violations(-104.0, 0.0) == 0
    assert oracle.lib().orc_exp_monotone_violations(0.0, 89.0) == 0


def test_uniform_normal_randint_moments(oracle):
    u = oracle.uniform(oracle.PRNGKey(2), (400000,))
    assert u.min() >= 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    z = oracle.normal(oracle.PRNGKey(1), (400000,))
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1) < 1e-2
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 5e-2
    r = oracle.randint(oracle.PRNGKey(3), (100000,), 0, 10)
    assert r.min() == 0 and r.max() == 9
    assert np.all(np.abs(np.bincount(r) / 1e5 - 0.1) < 5e-3)


def _assoc_scan_np(x):
    """lax.associative_scan(add) restated with numpy slices (the JAX source's own structure)."""
    x = np.asarray(x, np.float32)
    n = x.shape[0]
    if n < 2:
        return x.copy()
    red = (x[0:-1:2] + x[1::2]).astype(np.float32)
    odd = _assoc_scan_np(red)
    if n % 2 == 0:
        even = (odd[:-1] + x[2::2]).astype(np.float32)
    else:
        even = (odd + x[2::2]).astype(np.float32)
    even = np.concatenate([x[:1], even])
    out = np.empty(n, np.float32)
    out[0::2] = even
    out[1::2] = odd
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 13, 100, 255, 256, 257, 1000, 4097])
def test_cumsum_is_associative_scan_order(n, oracle):
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1, n).astype(np.float32)
    np.testing.assert_array_equal(oracle.cumsum(x).view(np.uint32), _assoc_scan_np(x).view(np.uint32))
    np.testing.assert_allclose(oracle.cumsum(x), np.cumsum(x.astype(np.float64)), rtol=1e-5)
    np.testing.assert_allclose(oracle.tree_sum(x), x.astype(np.float64).sum(), rtol=1e-5)


def test_searchsorted_matches_numpy_on_monotone_input(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 10, 1000):
        a = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
        q = np.concatenate([rng.uniform(-0.1, 1.1, 500).astype(np.float32), a[:20]])
        np.testing.assert_array_equal(oracle.searchsorted(a, q), np.searchsorted(a, q, side="left"))


def test_logsumexp_normalise(oracle):
    rng = np.random.default_rng(1)
    lw = rng.normal(0, 3, 5000).astype(np.float32)
    ref = np.log(np.sum(np.exp(lw.astype(np.float64) - lw.max()))) + lw.max()
    assert abs(float(oracle.logsumexp(lw)) - ref) < 1e-5 * abs(ref)
    w = oracle.normalise(lw, False)
    assert abs(w.astype(np.float64).sum() - 1) < 1e-5


# ---- restated from the reference's tests/test_cond_resamplings.py --------------------------------
def _cos_weights(n):
    w = np.cos(np.linspace(0, 2 * np.pi, n)) + 1
    return (w / w.sum()).astype(np.float32)


@pytest.mark.parametrize("name", ["cond_multinomial", "cond_killing"])
def test_unconditional_law(name, oracle):  # test_cond_resamplings.py:15-27
    n, nkeys = 1000, 4000
    w = _cos_weights(n)
    keys = oracle.split(oracle.PRNGKey(42), nkeys)
    counts = np.zeros(n)
    for k in keys:
        idx = getattr(oracle, name)(k, w, 0, 0, False)
        counts += np.bincount(idx[1:], minlength=n)
    np.testing.assert_allclose(counts / counts.sum(), w, atol=1.2e-3)


@pytest.mark.parametrize("name", ["stratified", "systematic", "multinomial", "killing"])
def test_unconditional_family_law(name, oracle):
    n, nkeys = 200, 3000
    w = _cos_weights(n)
    keys = oracle.split(oracle.PRNGKey(7), nkeys)
    counts = np.zeros(n)
    for k in keys:
        counts += np.bincount(getattr(oracle, name)(w, k), minlength=n)
    np.testing.assert_allclose(counts / counts.sum(), w, atol=1.5e-3)


@pytest.mark.parametrize("name", ["cond_multinomial", "cond_killing"])
@pytest.mark.parametrize("j", [0, 5, 50])
def test_conditional_bayes(name, j, oracle):  # test_cond_resamplings.py:33-53
    n, nkeys = 100, 20000
    w = _cos_weights(n)
    keys = oracle.split(oracle.PRNGKey(666), nkeys)
    counts = np.zeros(n)
    for k in keys:
        k1, k2 = oracle.split(k, 2)
        i = int(oracle.choice(k1, w, ()))
        idx = getattr(oracle, name)(k2, w, i, j, True)
        assert idx[j] == i
        counts += np.bincount(idx[1:], minlength=n)
    np.testing.assert_allclose(counts / counts.sum(), w, atol=1.5e-3)


def test_force_move_is_a_valid_kernel(oracle):
    """force_move leaves Cat(w) invariant (it is a Metropolised 'move away from k' kernel)."""
    rng = np.random.default_rng(3)
    w = rng.dirichlet(np.ones(8)).astype(np.float32)
    w = oracle.normalise(np.log(w), False)
    keys = oracle.split(oracle.PRNGKey(11), 40000)
    k = rng.choice(8, size=40000, p=w.astype(np.float64) / w.astype(np.float64).sum())
    out = np.array([oracle.force_move(key, w, int(kk))[0] for key, kk in zip(keys, k)])
    np.testing.assert_allclose(np.bincount(out, minlength=8) / 40000, w, atol=1e-2)


# ---- restated from the reference's tests/test_gibbs.py ------------------------------------------
def _toy_model(oracle, T, Tend):
    toy = toy_2d()
    return oracle.make_lg(toy["m0"], toy["cov0"], oracle.sde_const(-0.5, 1.0), np.linspace(0, Tend, T + 1), 1), toy


def test_gibbs_kernel_targets_the_posterior(oracle):  # test_gibbs.py:16-123
    m, toy = _toy_model(oracle, 100, 1.0)
    _, _, _, x0s = oracle.gibbs_chain_lg(m, oracle.PRNGKey(666), [0.0], toy["y0"], np.zeros(101, np.int32), 10, 10000)
    xs = x0s[10:, 0].astype(np.float64)
    np.testing.assert_allclose(xs.mean(), -1.8, rtol=5e-2)
    np.testing.assert_allclose(xs.var(), 1.68, rtol=2e-2)


@pytest.mark.parametrize("eb,ef", [(False, False), (True, True), (False, True)])
def test_gibbs_kernel_variants(eb, ef, oracle):
    # explicit_final draws u_0 ~ N(0, I): unbiased only once the forward SDE has mixed (T = 8)
    m, toy = _toy_model(oracle, 400, 8.0)
    _, _, _, x0s = oracle.gibbs_chain_lg(m, oracle.PRNGKey(1), [0.0], toy["y0"], np.zeros(401, np.int32), 10, 3000,
                                         explicit_backward=eb, explicit_final=ef)
    xs = x0s[10:, 0].astype(np.float64)
    assert abs(xs.mean() + 1.8) < 0.15
    assert abs(xs.var() - 1.68) < 0.25


def test_bootstrap_filter_and_smoother(oracle):
    """Averaged over forward y-paths drawn from y0, the filter's terminal particle cloud follows
    p(x0 | y0) = N(-1.8, 1.68) (tests/test_filters.py restated on the bridge model: the Kalman
    target is Gaussian conditioning), and the backward smoother's trajectory ends in the same law.
    (Particle 0 alone is NOT an exchangeable draw after stratified resampling, so the cloud
    average is used.)"""
    m, toy = _toy_model(oracle, 100, 4.0)
    means, second, sm = [], [], []
    keys = oracle.split(oracle.PRNGKey(5), 400)
    for key in keys:
        k1, k2, k3, k4 = oracle.split(key, 4)
        path = oracle.lg_fwd_sampler(m, k1, np.array([0.0, toy["y0"][0]], np.float32))
        vs = path[::-1, 1:].copy()
        init = oracle.normal(k2, (200, 1))
        filt, nell = oracle.bootstrap_filter_lg(m, k3, vs, init, "stratified", return_last=False)
        assert np.isfinite(nell)
        last = filt[-1, :, 0].astype(np.float64)
        means.append(last.mean())
        second.append((last ** 2).mean())
        sm.append(oracle.backward_smoother_lg(m, k4, filt, vs)[-1, 0])
    mean = np.mean(means)
    var = np.mean(second) - mean ** 2
    assert abs(mean + 1.8) < 0.15
    assert abs(var - 1.68) < 0.25
    assert abs(np.mean(sm) + 1.8) < 0.25


def test_pmcmc_filter_step_loglik_is_consistent(oracle):
    """log_ell of pmcmc_filter_step and -nell of bootstrap_filter estimate the same quantity."""
    m, toy = _toy_model(oracle, 40, 2.0)
    k1, k2, k3 = oracle.split(oracle.PRNGKey(3), 3)
    path = oracle.lg_fwd_sampler(m, k1, np.array([0.0, 0.0], np.float32))
    vs = path[::-1, 1:].copy()
    a, b = [], []
    for key in oracle.split(k2, 200):
        ka, kb, kc = oracle.split(key, 3)
        init = oracle.normal(ka, (300, 1))
        a.append(float(oracle.pmcmc_filter_step_lg(m, kb, vs, init)[1]))
        b.append(-float(oracle.bootstrap_filter_lg(m, kc, vs, init)[1]))
    assert abs(np.mean(a) - np.mean(b)) < 0.15


def test_backward_passes_return_consistent_trajectories(oracle):
    m, toy = _toy_model(oracle, 30, 1.0)
    rng = np.random.default_rng(0)
    key = oracle.PRNGKey(4)
    us_star = rng.normal(size=(31, 1)).astype(np.float32)
    vs = rng.normal(size=(31, 1)).astype(np.float32)
    bs = rng.integers(0, 20, 31).astype(np.int32)
    us0 = np.tile(us_star[0], (20, 1)).astype(np.float32)
    fp = oracle.csmc_forward_pass_lg(m, key, us_star, bs, vs, us0, np.full(20, -np.log(20), np.float32))
    # pinned reference particle at every step (csmc.py:143,152)
    for k in range(31):
        np.testing.assert_array_equal(fp["uss"][k, bs[k]], us_star[k])
    for k in range(30):
        assert fp["As"][k, bs[k + 1]] == bs[k]      # conditional resampling forces the ancestor
    xs, Bs = oracle.backward_scanning_pass(key, fp["As"], fp["uss"], fp["log_wss"][-1])
    for k in range(30, 0, -1):
        assert Bs[k - 1] == fp["As"][k - 1, Bs[k]]
        np.testing.assert_array_equal(xs[k], fp["uss"][k, Bs[k]])
    xs2, Bs2 = oracle.backward_sampling_pass_lg(m, key, vs, fp["uss"], fp["log_wss"])
    for k in range(31):
        np.testing.assert_array_equal(xs2[k], fp["uss"][k, Bs2[k]])


def test_openmp_build_is_bit_identical(oracle):
    """The OpenMP variant (cpu_baseline leg of bench.py) parallelises only independent particle
    loops: same bits as the serial build at any thread count."""
    m, toy = _toy_model(oracle, 20, 1.0)
    a, _ = oracle.bench_gibbs_lg(m, 7, [0.0], toy["y0"], 8192, 1, threads=1)
    b, used = oracle.bench_gibbs_lg(m, 7, [0.0], toy["y0"], 8192, 1, threads=4)
    assert used >= 1
    assert a.view(np.uint32) == b.view(np.uint32)


def test_ess_definition_against_float64():
    """orc_ess = 1 / sum w^2 of the normalised weights: against float64 numpy, and its two closed forms (uniform weights:
    n; one dominant weight: 1)."""
    import oracle as O
    rng = np.random.default_rng(7)
    for n in (1, 5, 1000, 70001):
        lw = rng.normal(0, 1.5, n).astype(np.float32)
        w = np.exp(lw.astype(np.float64) - np.logaddexp.reduce(lw.astype(np.float64)))
        assert abs(float(O.ess(lw)) - 1.0 / np.sum(w * w)) <= 2e-5 / np.sum(w * w)
        assert abs(float(O.ess(np.full(n, -3.0, np.float32))) - n) <= 1e-5 * n
    lw = np.full(100, -80.0, np.float32)
    lw[17] = 0.0
    assert float(O.ess(lw)) == 1.0
