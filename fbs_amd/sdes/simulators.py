"""SDE simulators with the reference's signatures (fbs/sdes/simulators.py).

``euler_maruyama`` / ``reverse_simulator`` / ``discrete_time_simulator`` take arbitrary drift
closures on torch tensors, so their time loop is host Python with the noise drawn by the JAX-
compatible device PRNG.  ``doob_bridge_simulator`` has a drift that is affine in x for every
scalar linear SDE, so the whole (T x integration_nsteps)-step path is ONE kernel launch
(fbsmi_affine_em_path).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib, ops
from .linear import LinearSDE, _as_np, _bridge_drift_coeffs


def reverse_simulator(key, u0, ts, score, drift, dispersion, integration_nsteps: int = 1,
                      integrator: str = 'euler-maruyama'):
    """Simulate the time-reversal of an SDE (fbs/sdes/simulators.py:8-50)."""
    T = float(_as_np(ts).reshape(-1)[-1])

    def reverse_drift(u, t):
        return -drift(u, T - t) + dispersion(T - t) ** 2 * score(u, T - t)

    def reverse_dispersion(t):
        return dispersion(T - t)

    if integrator == 'euler-maruyama':
        return euler_maruyama(key, u0, ts, reverse_drift, reverse_dispersion, integration_nsteps=integration_nsteps)
    raise NotImplementedError(f'Integrator {integrator} not implemented.')


def euler_maruyama(key, x0, ts, drift, dispersion, integration_nsteps: int = 1, return_path: bool = False):
    """Euler-Maruyama with sub-stepping (fbs/sdes/simulators.py:53-106)."""
    ts_np = _as_np(ts).reshape(-1)
    n = ts_np.size - 1
    if not isinstance(x0, torch.Tensor):
        x0 = torch.as_tensor(np.asarray(x0, np.float32), device=ops._default_device())
    ops._require_cuda(x0, "x0")
    keys = ops.split(key, n)
    x = x0.to(torch.float32).contiguous()
    path = [x]
    numel = x.numel()
    for k in range(n):
        t, t_next = float(ts_np[k]), float(ts_np[k + 1])
        ddt = abs(t_next - t) / integration_nsteps
        sub_ts = np.linspace(t, t_next - ddt, integration_nsteps)
        k0, k1 = int(keys[k][0]), int(keys[k][1])
        for j in range(integration_nsteps):
            t_ = float(sub_ts[j])
            # x + drift * ddt + dispersion * sqrt(ddt) * rnds[j] with rnds = normal(keys[k], (nsub, *shape)) (:91-99): one
            # kernel per sub-step, the noise drawn inside it (fbsmi_em_update)
            f = drift(x, t_)
            if not (isinstance(f, torch.Tensor) and f.shape == x.shape):
                f = torch.as_tensor(f, dtype=torch.float32, device=x.device).expand(x.shape)
            f = f.to(torch.float32).contiguous()
            c = float(np.float32(float(dispersion(t_)) * float(np.sqrt(ddt))))
            out = torch.empty_like(x)
            _lib.call("fbsmi_em_update", x.data_ptr(), f.data_ptr(), float(np.float32(ddt)), c, k0, k1,
                      integration_nsteps * numel, j * numel, numel, out.data_ptr(), ops._stream())
            x = out
        if return_path:
            path.append(x)
    return torch.stack(path, dim=0) if return_path else x


def discrete_time_simulator(key, x0, ts, f, q):
    """X(t_{k+1}) = f(X(t_k), t_{k+1}, t_k) + q(t_{k+1}, t_k) w  (fbs/sdes/simulators.py:109-123)."""
    ts_np = _as_np(ts).reshape(-1)
    if not isinstance(x0, torch.Tensor):
        x0 = torch.as_tensor(np.asarray(x0, np.float32), device=ops._default_device())
    rnds = ops.normal(key, (ts_np.size - 1,) + tuple(x0.shape), device=x0.device)
    x = x0
    for k in range(ts_np.size - 1):
        x = f(x, float(ts_np[k + 1]), float(ts_np[k])) + q(float(ts_np[k + 1]), float(ts_np[k])) * rnds[k]
    return x


def doob_bridge_simulator(key, sde: LinearSDE, x0, xT, ts, integration_nsteps: int = 1, replace: bool = False):
    """Doob h-transform bridge of a linear SDE from x0 to xT (fbs/sdes/simulators.py:126-160).

    bridge_drift(x, t) = A(t) x + B(t) xT is affine in x, so the path is one kernel launch: the
    host tabulates A, B and the dispersion at every sub-step time (float64 -> float32)."""
    ts_np = _as_np(ts).reshape(-1)
    T = ts_np.size - 1
    nsub = int(integration_nsteps)
    dev = x0.device if isinstance(x0, torch.Tensor) else ops._default_device()
    x0t = torch.as_tensor(np.asarray(x0, np.float32)) if not isinstance(x0, torch.Tensor) else x0
    xTt = torch.as_tensor(np.asarray(xT, np.float32)) if not isinstance(xT, torch.Tensor) else xT
    x0t = x0t.to(dev, torch.float32).contiguous()
    xTt = xTt.to(dev, torch.float32).contiguous()
    shape = tuple(x0t.shape)
    D = x0t.numel()
    Tend = float(ts_np[-1])
    A = np.zeros(T * nsub)
    B = np.zeros(T * nsub)
    S = np.zeros(T * nsub)
    ddt = np.zeros(T)
    for k in range(T):
        t, t_next = float(ts_np[k]), float(ts_np[k + 1])
        h = abs(t_next - t) / nsub
        ddt[k] = h
        for j, t_ in enumerate(np.linspace(t, t_next - h, nsub)):
            A[k * nsub + j], B[k * nsub + j] = _bridge_drift_coeffs(sde, float(t_), Tend)
            S[k * nsub + j] = float(sde.dispersion(float(t_)))
    keys = ops.split(key, T)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    keys_t = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    At, Bt, St, ht = up(A), up(B), up(S), up(ddt)
    out = torch.empty((T + 1, D), dtype=torch.float32, device=dev)
    _lib.call("fbsmi_affine_em_path", keys_t.data_ptr(), At.data_ptr(), Bt.data_ptr(), St.data_ptr(), ht.data_ptr(),
              xTt.reshape(-1).data_ptr(), x0t.reshape(-1).data_ptr(), T, nsub, D, int(bool(replace)), out.data_ptr(),
              ops._stream())
    return out.reshape((T + 1,) + shape)
