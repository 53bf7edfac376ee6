"""CPU: the tabulator statistics (fbs_amd/metrics.py) against closed forms and the reference's own tests
(tests/test_utils.py:9-13: KL and Bures vanish on identical Gaussians)."""
import numpy as np

from fbs_amd import metrics


def test_kl_and_bures_vanish_on_identical_gaussians():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(5, 5))
    cov = A @ A.T + np.eye(5)
    m = rng.normal(size=5)
    assert abs(metrics.kl(m, cov, m, cov)) < 1e-10
    assert abs(metrics.bures_dist(m, cov, m, cov)) < 1e-8


def test_closed_forms_in_one_dimension():
    m0, v0, m1, v1 = 0.3, 2.0, -0.5, 0.7
    want_kl2 = v0 / v1 - 1 + (m1 - m0) ** 2 / v1 + np.log(v1) - np.log(v0)       # twice KL(N0 || N1): the reference's convention
    assert abs(metrics.kl(np.array([m0]), np.array([[v0]]), np.array([m1]), np.array([[v1]])) - want_kl2) < 1e-12
    want_w2 = (m0 - m1) ** 2 + (np.sqrt(v0) - np.sqrt(v1)) ** 2
    assert abs(metrics.bures_dist(np.array([m0]), np.array([[v0]]), np.array([m1]), np.array([[v1]])) - want_w2) < 1e-12
    S = np.array([[2.0, 0.4], [0.4, 0.5]])
    R = metrics.sqrtm(S)
    np.testing.assert_allclose(R @ R, S, atol=1e-12)
    np.testing.assert_allclose(metrics.sqrtm(S, "schur"), R, atol=1e-10)


def test_error_statistics_of_exact_samples(tmp_path):
    rng = np.random.default_rng(1)
    d = 4
    A = rng.normal(size=(d, d))
    cov = A @ A.T / d + 0.5 * np.eye(d)
    mean = rng.normal(size=d)
    one = rng.multivariate_normal(mean, cov, size=200_000)
    s = metrics.toy_error_statistics(one, mean, cov)
    assert s["mean"] < 1e-2 and s["var"] < 2e-2 and s["kl"] < 1e-3 and s["bures"] < 1e-3 and s["skew"] < 2e-2 and s["kurt"] < 5e-2
    chains = rng.multivariate_normal(mean, cov, size=(3, 50_000))
    c = metrics.toy_error_statistics(chains, mean, cov)
    assert c["mean"] < 2e-2 and c["kl"] < 2e-3
    biased = metrics.toy_error_statistics(one + 0.5, mean, cov)
    assert abs(biased["mean"] - 0.5) < 1e-2 and biased["kl"] > 0.1
    for i in range(2):
        np.savez(tmp_path / f"run-{i}.npz", samples=one[i::2][:1000], gp_mean=mean, gp_cov=cov)
    t = metrics.tabulate(sorted(str(p) for p in tmp_path.glob("run-*.npz")))
    assert set(t) == {"mean", "var", "kl", "bures", "skew", "kurt"} and all(len(v) == 2 for v in t.values())
