"""Shared test inputs: the models of the reference's own tests."""
import numpy as np


def toy_2d():
    """2-D joint Gaussian of tests/test_gibbs.py:24-27 (du = dv = 1)."""
    return dict(m0=np.array([-1., 1.]), cov0=np.array([[2., 0.4], [0.4, 0.5]]), y0=np.array([0.], np.float32), du=1)


def toy_4d():
    """A 4-D joint Gaussian (du = dv = 2) to exercise the generic-dimension code paths."""
    rng = np.random.default_rng(7)
    A = rng.normal(size=(4, 4))
    cov0 = A @ A.T / 4 + 0.5 * np.eye(4)
    return dict(m0=np.array([0.5, -1., 1., 0.2]), cov0=cov0, y0=np.array([0.3, -0.4], np.float32), du=2)


def toy_31():
    """du = 3, dv = 1."""
    rng = np.random.default_rng(11)
    A = rng.normal(size=(4, 4))
    cov0 = A @ A.T / 4 + 0.5 * np.eye(4)
    return dict(m0=np.array([0.1, 0.2, -0.3, 1.]), cov0=cov0, y0=np.array([0.7], np.float32), du=3)


def toy_gp(d, dv=None, seed=5):
    """The d-dimensional Gaussian-process toy of experiments/toy/gp_gibbs.py:32-58 (Matern-1/2 covariance on
    linspace(0, 5, d), unit observation noise); with dv != d the observation is of the first dv coordinates."""
    dv = d if dv is None else dv
    zs = np.linspace(0., 5., d)
    cov = np.exp(-np.abs(zs[None, :] - zs[:, None]))
    H = np.eye(d)[:dv]
    joint = np.block([[cov, cov @ H.T], [H @ cov, H @ cov @ H.T + np.eye(dv)]])
    rng = np.random.default_rng(seed)
    return dict(m0=np.zeros(d + dv), cov0=joint, y0=rng.normal(size=dv).astype(np.float32), du=d)


def oracle_model_from(O, bridge):
    """An oracle LGModel fed with the PRODUCT's float32 tables (parity then isolates the kernels)."""
    h = bridge.host
    return O.LGModel(bridge.du, bridge.dv, bridge.dt, h["G"], h["g"], h["sd"], h["lognorm"], h["F"], h["sqQ"])


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a
