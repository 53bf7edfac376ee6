"""Scalar-coefficient linear SDEs and their exact discretisation.

Mirrors fbs/sdes/linear.py:9-227 of the reference (class and function names, argument orders).
Coefficient evaluations are scalar work and run on the host in float64 (numpy); anything that
touches particles or paths runs on the GPU through libfbsmi.
"""
from __future__ import annotations


import numpy as np
import torch

from .. import _lib, ops


class LinearSDE:
    pass


def _as_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().astype(np.float64)
    return np.asarray(x, dtype=np.float64)


def _bridge_drift_coeffs(sde, t, T):
    """bridge_drift(x, t, target, T) = A x + B target for a scalar linear SDE.

    The reference differentiates log N(target; F(T,t) x, Q(T,t)) w.r.t. x with jax.grad
    (fbs/sdes/linear.py:36-45, 83-92); in closed form the score is F (target - F x) / Q, hence
    A = a(t) - b(t)^2 F^2 / Q and B = b(t)^2 F / Q."""
    F, Q = discretise_linear_sde_np(sde, T, t)
    a = float(sde.drift(1.0, t))
    b2 = float(sde.dispersion(t)) ** 2
    return a - b2 * F * F / Q, b2 * F / Q


class StationaryConstLinearSDE(LinearSDE):
    """dX(t) = a X(t) dt + b dW(t), where `b^2 / a = 2 sigma^2`.  (fbs/sdes/linear.py:13-45)"""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def drift(self, x, t):
        return self.a * x

    def dispersion(self, t):
        return self.b

    def mean(self, t, s, m0):
        return m0 * np.exp(self.a * (_as_np(t) - _as_np(s)))

    def variance(self, t, s):
        return self.b ** 2 / (2 * self.a) * (np.exp(2 * self.a * (_as_np(t) - _as_np(s))) - 1)

    def bridge_drift(self, x, t, target, T):
        A, B = _bridge_drift_coeffs(self, float(t), float(T))
        return A * x + B * target


class StationaryLinLinearSDE(LinearSDE):
    r"""dX(t) = -0.5 \beta(t) X(t) dt + \sqrt{\beta(t)} dW(t), linear beta schedule.
    (fbs/sdes/linear.py:48-92)"""

    def __init__(self, beta_min, beta_max, t0, T):
        self.beta_min, self.beta_max, self.t0, self.T = beta_min, beta_max, t0, T

    def beta(self, t):
        beta_min, beta_max, t0, T = self.beta_min, self.beta_max, self.t0, self.T
        return (beta_max - beta_min) / (T - t0) * t + (beta_min * T - beta_max * t0) / (T - t0)

    def beta_integral(self, t, s):
        beta_min, beta_max, t0, T = self.beta_min, self.beta_max, self.t0, self.T
        return 0.5 * (t - s) * ((beta_max - beta_min) / (T - t0) * (t + s)
                                + 2 * (beta_min * T - beta_max * t0) / (T - t0))

    def drift(self, x, t):
        return -0.5 * self.beta(t) * x

    def dispersion(self, t):
        return np.sqrt(self.beta(t))

    def mean(self, t, s, m0):
        return m0 * np.exp(-0.5 * self.beta_integral(_as_np(t), _as_np(s)))

    def variance(self, t, s):
        return 1 - np.exp(-self.beta_integral(_as_np(t), _as_np(s)))

    def bridge_drift(self, x, t, target, T):
        A, B = _bridge_drift_coeffs(self, float(t), float(T))
        return A * x + B * target


class StationaryExpLinearSDE(LinearSDE):
    """dX(t) = a(t) X(t) dt + b(t) dW(t), a(t) = a exp(c (t - z)), b(t) = b exp(c (t - z) / 2).
    (fbs/sdes/linear.py:95-112)"""

    def __init__(self, a, b, c, z):
        self.a, self.b, self.c, self.z = a, b, c, z

    def drift(self, x, t):
        return self.a * np.exp(self.c * (t - self.z)) * x

    def dispersion(self, t):
        return self.b * np.exp(self.c * (t - self.z) / 2)


def discretise_linear_sde_np(sde: LinearSDE, t, s):
    """(F, Q) of x(t) | x(s) in float64 on the host: fbs/sdes/linear.py:169-184."""
    t, s = _as_np(t), _as_np(s)
    if isinstance(sde, StationaryLinLinearSDE):
        r = sde.beta_integral(t, s)
        return np.exp(-0.5 * r), 1 - np.exp(-r)
    if isinstance(sde, StationaryConstLinearSDE):
        a, b = sde.a, sde.b
        return np.exp(a * (t - s)), b ** 2 / (2 * a) * (np.exp(2 * a * (t - s)) - 1)
    if isinstance(sde, StationaryExpLinearSDE):
        a, b, c, z = sde.a, sde.b, sde.c, sde.z
        stationary_variance = -b ** 2 / (2 * a)
        r = a * (np.exp(c * (t - z)) - np.exp(c * (s - z))) / c
        return np.exp(r), stationary_variance * (1 - np.exp(2 * r))
    raise NotImplementedError('...')


def _linear_path(F, S, x0: torch.Tensor, xi: torch.Tensor) -> torch.Tensor:
    """out[0] = x0, out[k+1] = F[k] out[k] + S[k] xi[k] on the device (fbsmi_linear_path)."""
    dev = x0.device
    T = int(F.shape[0])
    Ft = torch.from_numpy(np.ascontiguousarray(F, np.float32)).to(dev)
    St = torch.from_numpy(np.ascontiguousarray(S, np.float32)).to(dev)
    flat = x0.to(torch.float32).contiguous().reshape(-1)
    D = flat.numel()
    xi = xi.to(torch.float32).contiguous()
    out = torch.empty((T + 1, D), dtype=torch.float32, device=dev)
    _lib.call("fbsmi_linear_path", Ft.data_ptr(), St.data_ptr(), flat.data_ptr(), xi.data_ptr(), T, D, out.data_ptr(),
              ops._stream())
    return out.reshape((T + 1,) + tuple(x0.shape))


def make_linear_sde(sde: LinearSDE):
    """(discretise_linear_sde, cond_score_t_0, simulate_cond_forward): fbs/sdes/linear.py:165-227."""

    def discretise_linear_sde(t, s):
        return discretise_linear_sde_np(sde, t, s)

    def cond_score_t_0(x, t, x0, s):
        F, Q = discretise_linear_sde(t, s)
        return -(x - float(F) * x0) / float(Q)

    def simulate_cond_forward(key, x0, ts, t0: float = None, keep_path: bool = True):
        ts_np = _as_np(ts).reshape(-1)
        if not isinstance(x0, torch.Tensor):
            x0 = torch.as_tensor(np.asarray(x0, np.float32), device=ops._default_device())
        ops._require_cuda(x0, "x0")
        if keep_path:
            F, Q = discretise_linear_sde(ts_np[1:], ts_np[:-1])
            rnds = ops.normal(key, (ts_np.size - 1, x0.numel()), device=x0.device)
            return _linear_path(F, np.sqrt(Q), x0, rnds)
        Fs, Qs = discretise_linear_sde(ts_np, ts_np[0] if t0 is None else t0)
        rnds = ops.normal(key, (ts_np.size,) + tuple(x0.shape), device=x0.device)
        shp = (-1,) + (1,) * x0.dim()
        Fs_t = torch.as_tensor(np.asarray(Fs, np.float32), device=x0.device).reshape(shp)
        Qs_t = torch.as_tensor(np.sqrt(np.asarray(Qs)).astype(np.float32), device=x0.device).reshape(shp)
        return Fs_t * x0 + Qs_t * rnds

    return discretise_linear_sde, cond_score_t_0, simulate_cond_forward


def make_ou_sde(a, b):
    """Independent OU SDEs dX = a X dt + b dW (fbs/sdes/linear.py:115-162); same engine as
    make_linear_sde(StationaryConstLinearSDE(a, b)) -- the reference tests them bit-equal
    (tests/test_sdes.py:135-163)."""
    sde = StationaryConstLinearSDE(a, b)
    disc, score, sim = make_linear_sde(sde)

    def discretise_ou_sde(t):
        return disc(t, 0.)

    def cond_score_t_0(x, t, x0):
        return score(x, t, x0, 0.)

    def simulate_cond_forward(key, x0, ts, keep_path: bool = True):
        return sim(key, x0, ts, t0=0., keep_path=keep_path)

    return discretise_ou_sde, cond_score_t_0, simulate_cond_forward


def _sqrtm(mat):
    """Symmetric PSD matrix square root by eigen-decomposition (fbs/utils.py:24-31), float64 host."""
    vals, vecs = np.linalg.eigh(np.asarray(mat, np.float64))
    return vecs @ np.diag(np.sqrt(vals)) @ vecs.T


def make_gaussian_bw_sb(mean0, cov0, mean1, cov1, sig: float = 1.):
    """Gaussian Schrodinger bridge with a Brownian reference on [0, 1] (fbs/sdes/linear.py:397-457;
    Table 1 of "The Schrodinger Bridge between Gaussian Measures has a Closed Form", 2023).

    Returns (marginal_mean(t), marginal_cov(t), drift(x, t)).  The matrices are host float64; `drift`
    applies the affine map s(t)^T cov_t^{-1} (x - m_t) - mean0 + mean1 to a batch of states x
    (..., d) on whatever device x lives."""
    mean0, mean1 = np.asarray(_as_np(mean0)), np.asarray(_as_np(mean1))
    cov0, cov1 = np.asarray(_as_np(cov0)), np.asarray(_as_np(cov1))
    d = mean0.shape[0]
    eye = np.eye(d)
    sqrt0 = _sqrtm(cov0)
    D_sig = _sqrtm(4 * sqrt0 @ cov1 @ sqrt0 + sig ** 4 * eye)
    C_sig = 0.5 * (sqrt0 @ np.linalg.solve(sqrt0.T, D_sig.T).T - sig ** 2 * eye)

    def marginal_mean(t):
        return (1 - t) * mean0 + t * mean1

    def marginal_cov(t):
        return (1 - t) ** 2 * cov0 + t ** 2 * cov1 + t * (1 - t) * (C_sig + C_sig.T) + t * sig ** 2 * (1 - t) * eye

    def s(t):
        pt = t * cov1 + (1 - t) * C_sig
        qt = (1 - t) * cov0 + t * C_sig
        return pt - qt.T - sig ** 2 * t * eye

    def drift(x, t):
        t = float(t)
        M = s(t).T @ np.linalg.inv(marginal_cov(t))          # (d, d)
        const = mean1 - mean0 - M @ marginal_mean(t)
        if isinstance(x, torch.Tensor):
            Mt = torch.as_tensor(M, dtype=x.dtype, device=x.device)
            return x @ Mt.T + torch.as_tensor(const, dtype=x.dtype, device=x.device)
        return np.asarray(x) @ M.T + const

    return marginal_mean, marginal_cov, drift
