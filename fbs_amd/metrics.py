"""Sample-quality statistics of the toy experiments (fbs/utils.py:24-53, experiments/tabulators/tabulate_toy.py:38-62):
what the reference's tabulator computes from the `.npz` files its drivers -- and this package's examples/toy_*.py, which
write the same schema (samples, gp_mean, gp_cov) -- leave behind.  Offline post-processing on the host: float64 numpy."""
from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.stats


def sqrtm(mat, method: str = "eigh"):
    """Matrix (Hermite) square root (fbs/utils.py:24-31)."""
    mat = np.asarray(mat, np.float64)
    if method == "eigh":
        vals, vecs = np.linalg.eigh(mat)
        return vecs @ np.diag(np.sqrt(vals)) @ vecs.T
    return np.real(scipy.linalg.sqrtm(mat))


def bures_dist(m0, cov0, m1, cov1):
    """The (squared) Wasserstein-2 distance between two Gaussians (fbs/utils.py:34-39)."""
    m0, m1 = np.asarray(m0, np.float64), np.asarray(m1, np.float64)
    cov0, cov1 = np.asarray(cov0, np.float64), np.asarray(cov1, np.float64)
    s = sqrtm(cov0)
    A = cov0 + cov1 - 2 * sqrtm(s @ cov1 @ s)
    return float(np.sum((m0 - m1) ** 2) + np.trace(A))


def kl(m0, cov0, m1, cov1):
    """The reference's `kl` (fbs/utils.py:46-53), verbatim in its convention: tr(S1^-1 S0) - d + (m1-m0)' S1^-1 (m1-m0)
    + log det S1 - log det S0, i.e. TWICE the Kullback-Leibler divergence KL(N0 || N1)."""
    m0, m1 = np.asarray(m0, np.float64), np.asarray(m1, np.float64)
    cov0, cov1 = np.asarray(cov0, np.float64), np.asarray(cov1, np.float64)
    d = m0.shape[-1]
    c0, c1 = scipy.linalg.cho_factor(cov0), scipy.linalg.cho_factor(cov1)
    logdet = lambda c: 2 * np.sum(np.log(np.abs(np.diag(c[0]))))
    dm = m1 - m0
    return float(np.trace(scipy.linalg.cho_solve(c1, cov0)) - d + dm @ scipy.linalg.cho_solve(c1, dm) + logdet(c1) - logdet(c0))


def toy_error_statistics(samples, gp_mean, gp_cov):
    """One Monte-Carlo run of the toy experiment -> the six error figures of tabulate_toy.py:38-62.

    samples (nsamples, d): one sample set (filter / twisted / csgm); samples (nchains, nsamples, d): MCMC chains
    (gibbs / pmcmc), whose per-chain errors are averaged."""
    samples = np.asarray(samples, np.float64)
    gp_mean, gp_cov = np.asarray(gp_mean, np.float64), np.asarray(gp_cov, np.float64)
    if samples.ndim == 3:                                                          # :44-54
        means = samples.mean(axis=1)
        covs = np.stack([np.cov(s, rowvar=False) for s in samples])
        return dict(mean=float(np.mean(np.abs(means - gp_mean[None]))),
                    var=float(np.mean(np.abs(np.diagonal(covs - gp_cov[None], axis1=1, axis2=2)))),
                    kl=float(np.mean([kl(gp_mean, gp_cov, m, c) for m, c in zip(means, covs)])),
                    bures=float(np.mean([bures_dist(gp_mean, gp_cov, m, c) for m, c in zip(means, covs)])),
                    skew=float(np.mean(np.abs(scipy.stats.skew(samples, axis=1)))),
                    kurt=float(np.mean(np.abs(scipy.stats.kurtosis(samples, axis=1, fisher=True)))))
    mean, cov = samples.mean(axis=0), np.cov(samples, rowvar=False)                # :55-63
    return dict(mean=float(np.mean(np.abs(mean - gp_mean))), var=float(np.mean(np.abs(np.diag(cov) - np.diag(gp_cov)))),
                kl=kl(gp_mean, gp_cov, mean, cov), bures=bures_dist(gp_mean, gp_cov, mean, cov),
                skew=float(np.mean(np.abs(scipy.stats.skew(samples, axis=0)))),
                kurt=float(np.mean(np.abs(scipy.stats.kurtosis(samples, axis=0, fisher=True)))))


def tabulate(files):
    """Mean and standard deviation over Monte-Carlo runs of every figure (the print of tabulate_toy.py:72-76).
    files: iterable of `.npz` paths with `samples`, `gp_mean`, `gp_cov`."""
    rows = []
    for f in files:
        r = np.load(f)
        rows.append(toy_error_statistics(r["samples"], r["gp_mean"], r["gp_cov"]))
    return {k: (float(np.mean([r[k] for r in rows])), float(np.std([r[k] for r in rows]))) for k in rows[0]}
