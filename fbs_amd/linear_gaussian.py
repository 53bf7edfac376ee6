"""Linear-Gaussian bridge model: the analytic-score toy of the reference as a model *descriptor*.

For a scalar-coefficient linear SDE ``dZ = a(t) Z dt + b(t) dW`` on the joint ``Z = (X, Y)`` with a
Gaussian prior ``N(m0, cov0)``, the marginal at forward time t is ``N(F m0, F^2 cov0 + Q I)``
(experiments/toy/gp_gibbs.py:73-75), the score is ``-cov_t^{-1}(z - m_t)`` (:78-81) and the
reverse-time drift (:94-95) is affine in z:  ``f(z, tau) = G z + g`` with
``G = -a I - b^2 cov^{-1}``, ``g = b^2 cov^{-1} m`` at forward time ``T - tau``.

The descriptor precomputes the per-step tables in float64 on the host, keeps float32 copies on
the GPU, and exposes

* the reference's closures (``transition_sampler``, ``transition_logpdf``, ``likelihood_logpdf``,
  ``fwd_sampler``, ``fwd_ys_sampler``, ``unpack``, ``ref_sampler``) with the reference's
  signatures, each one a HIP kernel launch;
* the fused whole-sweep engine (``gibbs_kernel`` / ``gibbs_chain``), which
  ``fbs_amd.samplers.gibbs_kernel`` dispatches to when it is handed these closures.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import numpy as np
import torch

from . import _lib, ops
from .sdes.linear import LinearSDE, discretise_linear_sde_np


def lg_tables(m0, cov0, sde: LinearSDE, ts, du: int, dt: Optional[float] = None) -> dict:
    """Float64 per-step tables; step k is the reverse-time interval starting at t_prev = ts[k]."""
    m0 = np.asarray(m0, np.float64).reshape(-1)
    cov0 = np.asarray(cov0, np.float64)
    ts = np.asarray(ts, np.float64).reshape(-1)
    D, T = m0.size, ts.size - 1
    Tend = ts[-1]
    dt = float((Tend - ts[0]) / T) if dt is None else float(dt)  # gp_gibbs.py:63 (constant T / nsteps)
    G = np.zeros((T, D, D))
    g = np.zeros((T, D))
    sd = np.zeros(T)
    F = np.zeros(T)
    sqQ = np.zeros(T)
    eye = np.eye(D)
    for k in range(T):
        t_fwd = Tend - ts[k]
        Ft, Qt = discretise_linear_sde_np(sde, t_fwd, ts[0])
        P = np.linalg.inv(Ft ** 2 * cov0 + Qt * eye)
        a_t = float(sde.drift(1.0, t_fwd))       # a(t): drift(x, t) = a(t) x
        b_t = float(sde.dispersion(t_fwd))
        G[k] = -a_t * eye - b_t ** 2 * P
        g[k] = b_t ** 2 * (P @ (Ft * m0))
        sd[k] = np.sqrt(dt) * b_t
        Fk, Qk = discretise_linear_sde_np(sde, ts[k + 1], ts[k])
        F[k], sqQ[k] = Fk, np.sqrt(Qk)
    return dict(du=int(du), dv=int(D - du), dt=dt, G=G, g=g, sd=sd, lognorm=np.log(2 * np.pi * sd ** 2), F=F,
                sqQ=sqQ)


class _Closure:
    """A callable that remembers the model it came from (how gibbs_kernel recognises the fused path)."""

    def __init__(self, model, fn, role):
        self._fbsmi_lg = model
        self._role = role
        self._fn = fn
        self.__name__ = role

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)


class LinearGaussianBridge:
    def __init__(self, m0, cov0, sde: LinearSDE, ts, du: int, device=None, dt: Optional[float] = None):
        self.device = torch.device(device) if device is not None else ops._default_device()
        self.sde = sde
        self.ts_np = np.asarray(ts, np.float64).reshape(-1)
        self.m0 = np.asarray(m0, np.float64).reshape(-1)
        self.cov0 = np.asarray(cov0, np.float64)
        tab = lg_tables(m0, cov0, sde, ts, du, dt)
        self.tables64 = tab
        self.du, self.dv = tab["du"], tab["dv"]
        self.D = self.du + self.dv
        self.T = self.ts_np.size - 1
        self.dt = np.float32(tab["dt"])
        f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
        self.host = {k: f32(tab[k]) for k in ("G", "g", "sd", "lognorm", "F", "sqQ")}
        self.dev = {k: torch.from_numpy(v).to(self.device) for k, v in self.host.items()}
        self.struct = _lib.LGModelStruct(self.du, self.dv, self.T, float(self.dt), *(self.dev[k].data_ptr() for k in
                                         ("G", "g", "sd", "lognorm", "F", "sqQ")))
        self._sweeps = {}
        # closures with the reference's signatures
        self.transition_sampler = _Closure(self, self._transition_sampler, "transition_sampler")
        self.transition_logpdf = _Closure(self, self._transition_logpdf, "transition_logpdf")
        self.likelihood_logpdf = _Closure(self, self._likelihood_logpdf, "likelihood_logpdf")
        self.fwd_sampler = _Closure(self, self._fwd_sampler, "fwd_sampler")
        self.fwd_ys_sampler = _Closure(self, self._fwd_ys_sampler, "fwd_ys_sampler")
        self.unpack = _Closure(self, self._unpack, "unpack")
        self.ref_sampler = _Closure(self, self._ref_sampler, "ref_sampler")

    # -- helpers ---------------------------------------------------------------------------------
    def step_of(self, t_prev) -> int:
        """Index of the step that leaves time t_prev; the closures are tabulated on the bridge's own grid."""
        t = float(t_prev)
        k = int(np.argmin(np.abs(self.ts_np[:-1] - t)))
        if abs(self.ts_np[k] - t) > 1e-6 * max(1.0, abs(self.ts_np[-1])):
            raise ValueError(f"t_prev = {t} is not a point of this bridge's time grid (nearest: {self.ts_np[k]})")
        return k

    def same_grid(self, ts) -> bool:
        ts = np.asarray(ts.detach().cpu() if isinstance(ts, torch.Tensor) else ts, np.float64).reshape(-1)
        return ts.shape == self.ts_np.shape and bool(np.allclose(ts, self.ts_np, rtol=0.0, atol=1e-9 * max(1.0, abs(self.ts_np[-1]))))

    def _t(self, x, shape=None) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x, np.float32))
        x = x.to(self.device, torch.float32).contiguous()
        return x.reshape(shape) if shape is not None else x

    def _ref(self):
        return C.byref(self.struct)

    # -- closures (experiments/toy/gp_gibbs.py:109-149) -------------------------------------------
    def _unpack(self, xy):
        return xy[..., :self.du], xy[..., self.du:]

    def _transition_sampler(self, us_prev, v_prev, t_prev, key, row_slice=None):
        """row_slice = (offset, count, total): this call propagates rows [offset, offset+count) of an
        ensemble of `total` rows (sharded ensembles); the noise is that slice of the global draw."""
        k = self.step_of(t_prev)
        up = self._t(us_prev).reshape(-1, self.du)
        vp = self._t(v_prev, (self.dv,))
        out = torch.empty_like(up)
        k0, k1 = ops._k(key)
        if row_slice is None:
            _lib.call("fbsmi_lg_transition_sampler", self._ref(), k, float(self.host["sd"][k]),
                      float(self.host["lognorm"][k]), up.data_ptr(), vp.data_ptr(), k0, k1, up.shape[0],
                      out.data_ptr(), ops._stream())
        else:
            offset, count, total = (int(x) for x in row_slice)
            _lib.call("fbsmi_lg_transition_sampler_rows", self._ref(), k, float(self.host["sd"][k]),
                      float(self.host["lognorm"][k]), up.data_ptr(), vp.data_ptr(), k0, k1, total, offset, count,
                      out.data_ptr(), ops._stream())
        return out.reshape(us_prev.shape)

    def _logpdf(self, name, target, us_prev, v_prev, t_prev, dim):
        k = self.step_of(t_prev)
        up = self._t(us_prev).reshape(-1, self.du)
        vp = self._t(v_prev, (self.dv,))
        tg = self._t(target, (dim,))
        out = torch.empty(up.shape[0], dtype=torch.float32, device=self.device)
        _lib.call(name, self._ref(), k, float(self.host["sd"][k]), float(self.host["lognorm"][k]), tg.data_ptr(),
                  up.data_ptr(), vp.data_ptr(), up.shape[0], out.data_ptr(), ops._stream())
        return out

    def _likelihood_logpdf(self, v, us_prev, v_prev, t_prev):
        return self._logpdf("fbsmi_lg_likelihood_logpdf", v, us_prev, v_prev, t_prev, self.dv)

    def _transition_logpdf(self, u, us_prev, v_prev, t_prev):
        return self._logpdf("fbsmi_lg_transition_logpdf", u, us_prev, v_prev, t_prev, self.du)

    def _linear_path(self, key, z0):
        z0 = self._t(z0).reshape(-1)
        Dz = z0.numel()
        xi = ops.normal(key, (self.T, Dz), device=self.device)
        out = torch.empty((self.T + 1, Dz), dtype=torch.float32, device=self.device)
        _lib.call("fbsmi_linear_path", self.dev["F"].data_ptr(), self.dev["sqQ"].data_ptr(), z0.data_ptr(),
                  xi.data_ptr(), self.T, Dz, out.data_ptr(), ops._stream())
        return out

    def _fwd_sampler(self, key, x0, y0):  # gp_gibbs.py:144-145
        return self._linear_path(key, torch.cat([self._t(x0).reshape(-1), self._t(y0).reshape(-1)]))

    def _fwd_ys_sampler(self, key, y0):  # gp_gibbs.py:148-149
        return self._linear_path(key, self._t(y0).reshape(-1))

    def terminal_moments(self):
        """m_ref, cov_ref = forward_m_cov(T), gp_gibbs.py:84-86 (float64, host)."""
        Ft, Qt = discretise_linear_sde_np(self.sde, self.ts_np[-1], self.ts_np[0])
        return Ft * self.m0, Ft ** 2 * self.cov0 + Qt * np.eye(self.D)

    def _ref_sampler(self, key, yT, nsamples):  # gp_gibbs.py:138-141, p(u0 | v0) at the terminal time
        m_ref, cov_ref = self.terminal_moments()
        d = self.du
        yT = np.asarray(yT.detach().cpu() if isinstance(yT, torch.Tensor) else yT, np.float64).reshape(-1)
        Kyy = cov_ref[d:, d:]
        gain = cov_ref[:d, d:] @ np.linalg.inv(Kyy)
        m_ = m_ref[:d] + gain @ (yT - m_ref[d:])
        cov_ = cov_ref[:d, :d] - gain @ cov_ref[d:, :d]
        chol = np.linalg.cholesky(cov_)
        z = ops.normal(key, (int(nsamples), d), device=self.device)
        # jax: m_ + normal @ cholesky(cov_) -- the reference multiplies by the LOWER factor on the right
        return self._t(m_) + z @ self._t(chol)

    # -- fused engine ----------------------------------------------------------------------------
    def fused_sweep_supported(self, nparticles: int, explicit_final: bool = False) -> bool:
        """What fbsmi_lg_sweep_create accepts: du, dv <= 16 at any ensemble size up to 4M particles; du, dv <= 128
        (drift on the matrix cores) with at most 131072 slots (particles + 1 with explicit_final)."""
        wide = max(self.du, self.dv) > 16
        if not wide:
            return True
        return max(self.du, self.dv) <= 128 and nparticles + (1 if explicit_final else 0) <= 131072

    def fused_filter_supported(self, nparticles: int) -> bool:
        """What fbsmi_lg_filter_create accepts: du, dv <= 16, or du, dv <= 128 with at most 131072 particles."""
        if max(self.du, self.dv) <= 16:
            return True
        return max(self.du, self.dv) <= 128 and nparticles <= 131072

    def sweep_handle(self, nparticles: int, explicit_backward=True, explicit_final=False, store_path=None,
                     nchains: int = 1):
        store = (not explicit_backward) if store_path is None else bool(store_path)
        keyt = (int(nparticles), bool(explicit_backward), bool(explicit_final), store, int(nchains))
        h = self._sweeps.get(keyt)
        if h is None:
            h = LGSweep(self, *keyt)
            self._sweeps[keyt] = h
        return h

    def filter_handle(self, nparticles: int, flow: str, resampling: str = "stratified", store_path: bool = False):
        keyt = ("filter", int(nparticles), flow, resampling, bool(store_path))
        h = self._sweeps.get(keyt)
        if h is None:
            h = LGFilter(self, int(nparticles), flow, resampling, bool(store_path))
            self._sweeps[keyt] = h
        return h

    def gibbs_kernel(self, key, x0, y0, bs_star, nparticles, explicit_backward=True, explicit_final=False,
                     use_graph=True):
        """One fused sweep; same returns as fbs.samplers.gibbs_kernel: (x0, us_star, bs_star, acc)."""
        h = self.sweep_handle(nparticles, explicit_backward, explicit_final)
        return h.sweep(key, x0, y0, bs_star, use_graph=use_graph)


class LGSweep:
    """Owns one fbsmi_lg_sweep handle (device buffers + captured hipGraph) for `nchains` chains.

    With nchains == 1 the chain axis is squeezed from inputs and outputs (the reference's plain
    gibbs_kernel); with nchains > 1 every per-chain array carries a leading axis of that size, like
    the reference's jax.vmap(gibbs_kernel, in_axes=[0, 0, None, 0, 0]) (gp_gibbs.py:173)."""

    def __init__(self, model: LinearGaussianBridge, nparticles, eb, ef, store, nchains=1, _group=None):
        self.model = model
        self.nparticles, self.eb, self.ef, self.store, self.C = nparticles, eb, ef, store, int(nchains)
        self.n_rows = nparticles + 1 if ef else nparticles
        self.children, self.h = [], None
        # A batch of four or more chains is driven as two handles of half the chains each, on their own streams: the step
        # kernels are latency-bound, so the two halves' launches interleave and finish sooner than one full-size batch
        # (same results bit for bit; FBSMI_CHAIN_GROUPS=1 keeps one handle, =k asks for k groups).
        sz = None
        if _group is None:
            # Two groups from four chains, three from five (narrow models), except very large wide ensembles (launches of hundreds
            # of microseconds), which take four.  A FOURTH concurrent graph stream of short dependent launches is pathological
            # (round 3: 28-31 us per step with four groups of one chain against 16-17 with two or three groups, for the narrow toy
            # and for the d = 100 toy at 100 particles alike; d = 100 at 10 000 particles anywhere between 17 and 28 ms per sweep
            # with four groups against a steady 19-21 with two; at 100 000 particles four groups won on every box, 152 against
            # 174 ms), three are not: config 2's model with 8 chains as (3, 3, 2) 22.3 us per step against 23.9 as (4, 4), 16
            # chains as (6, 5, 5) 29.8 against 31.6, 5 chains as (2, 2, 1) 17.9 against 18.6 as (3, 2); at the reference's four
            # chains (2, 2) and (2, 1, 1) measure the same.  The host is not the limit (0.3 us per graph node); the queues are.
            # FBSMI_CHAIN_GROUPS=1 keeps one handle, =k asks for k groups, FBSMI_CHAIN_GROUP_SIZES=a,b,.. for explicit sizes.
            wide = max(model.du, model.dv) > 16
            sizes = os.environ.get("FBSMI_CHAIN_GROUP_SIZES")
            if sizes:
                sz = [int(x) for x in sizes.split(",")]
                if sum(sz) != self.C or any(x < 1 for x in sz):
                    sz = None          # (the variable is process-wide: batches of another size keep the default policy)
            if sz is None:
                G = int(os.environ.get("FBSMI_CHAIN_GROUPS", "0"))
                if G < 1:
                    if wide:
                        G = (4 if (self.C % 4 == 0 and nparticles >= 32768) else 2) if self.C >= 4 else 1
                    else:
                        G = 3 if self.C >= 5 else (2 if self.C >= 4 else 1)
                G = min(G, self.C)
                sz = [self.C // G + (1 if g < self.C % G else 0) for g in range(G)]
        if sz is not None and len(sz) > 1:
            first = np.cumsum([0] + sz[:-1])
            self.children = [LGSweep(model, nparticles, eb, ef, store, n, _group=(self.C, int(f))) for n, f in zip(sz, first)]
            self._harr = (C.c_void_p * len(sz))(*[c.h for c in self.children])
            return
        h = C.c_void_p()
        with torch.cuda.device(model.device):
            _lib.call("fbsmi_lg_sweep_create", C.byref(model.struct), nparticles, int(eb), int(ef), int(store),
                      self.C, C.byref(h))
        self.h = h
        if _group is not None:
            _lib.call("fbsmi_lg_sweep_set_group", self.h, int(_group[0]), int(_group[1]))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().fbsmi_lg_sweep_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _dev(self, x, dtype, shape):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        return x.to(self.model.device, dtype).contiguous().reshape(shape)

    def _key_t(self, key, n):
        k = np.asarray(key.detach().cpu() if isinstance(key, torch.Tensor) else key).astype(np.uint32).reshape(n, 2)
        return torch.from_numpy(k.view(np.int32).copy()).to(self.model.device)

    def _sq(self, t):
        return t[0] if self.C == 1 else t

    def sweep(self, key, x0, y0, bs_star, use_graph=True):
        """key (C,2) [or (2,)], x0 (C,du), y0 (dv,), bs_star (C,T+1) -> (x0, us_star, bs_star, acc)."""
        m, Cn = self.model, self.C
        if self.children:     # every group sweeps its chains (explicit per-chain keys: nothing to coordinate)
            k2 = np.asarray(key.detach().cpu() if isinstance(key, torch.Tensor) else key).reshape(Cn, 2)
            x2 = self._dev(x0, torch.float32, (Cn, m.du))
            b2 = self._dev(bs_star, torch.int32, (Cn, m.T + 1))
            outs, c0 = [], 0
            for ch in self.children:
                o = ch.sweep(k2[c0:c0 + ch.C], x2[c0:c0 + ch.C], y0, b2[c0:c0 + ch.C], use_graph=use_graph)
                outs.append([t.reshape((ch.C,) + tuple(t.shape[(0 if ch.C == 1 else 1):])) for t in o])
                c0 += ch.C
            return tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(4))
        kt = self._key_t(key, Cn)
        x0t = self._dev(x0, torch.float32, (Cn, m.du))
        y0t = self._dev(y0, torch.float32, (m.dv,))
        bst = self._dev(bs_star, torch.int32, (Cn, m.T + 1))
        x0n = torch.empty((Cn, m.du), dtype=torch.float32, device=m.device)
        usn = torch.empty((Cn, m.T + 1, m.du), dtype=torch.float32, device=m.device)
        bsn = torch.empty((Cn, m.T + 1), dtype=torch.int32, device=m.device)
        acc = torch.empty((Cn, m.T + 1), dtype=torch.uint8, device=m.device)
        _lib.call("fbsmi_lg_gibbs_sweep", self.h, kt.data_ptr(), x0t.data_ptr(), y0t.data_ptr(), bst.data_ptr(),
                  x0n.data_ptr(), usn.data_ptr(), bsn.data_ptr(), acc.data_ptr(), int(bool(use_graph)), ops._stream())
        return self._sq(x0n), self._sq(usn), self._sq(bsn), self._sq(acc.bool())

    def chain(self, key, x0, y0, bs_star, nsweeps, keep=True, use_graph=True):
        """nsweeps sweeps; per sweep ``key, subkey = split(key)``; one chain sweeps with subkey
        (tests/test_gibbs.py:115-118), C > 1 chains with split(subkey, C)[c] (gp_gibbs.py:183-185).
        Returns (key, x0, bs_star, x0s) with x0s of shape (nsweeps, [C,] du)."""
        m, Cn = self.model, self.C
        kt = self._key_t(key, 1)
        x0t = self._dev(x0, torch.float32, (Cn, m.du)).clone()
        y0t = self._dev(y0, torch.float32, (m.dv,))
        bst = self._dev(bs_star, torch.int32, (Cn, m.T + 1)).clone()
        x0s = torch.empty((nsweeps, Cn, m.du), dtype=torch.float32, device=m.device) if keep else None
        if self.children:
            _lib.call("fbsmi_lg_gibbs_chain_groups", self._harr, len(self.children), kt.data_ptr(), x0t.data_ptr(), y0t.data_ptr(),
                      bst.data_ptr(), int(nsweeps), x0s.data_ptr() if keep else None, int(bool(use_graph)), ops._stream())
        else:
            _lib.call("fbsmi_lg_gibbs_chain", self.h, kt.data_ptr(), x0t.data_ptr(), y0t.data_ptr(), bst.data_ptr(),
                      int(nsweeps), x0s.data_ptr() if keep else None, int(bool(use_graph)), ops._stream())
        key_out = kt.cpu().numpy().view(np.uint32).reshape(2).copy()
        if keep and Cn == 1:
            x0s = x0s[:, 0]
        return key_out, self._sq(x0t), self._sq(bst), x0s

    def views(self):
        """Parity views of the last sweep's CSMC forward pass (copies; leading chain axis if C > 1)."""
        m = self.model
        N, Cn = self.n_rows, self.C
        if self.children:
            parts = [ch.views() for ch in self.children]
            cat = lambda name: None if parts[0][name] is None else torch.cat(
                [p[name].reshape((ch.C,) + tuple(p[name].shape[(0 if ch.C == 1 else 1):])) for p, ch in zip(parts, self.children)], dim=0)
            return {name: cat(name) for name in parts[0]}
        spec = {"us_T": (0, torch.float32, (N, m.du)), "lw_T": (1, torch.float32, (N,)),
                "As": (2, torch.int32, (m.T, N)), "uss": (3, torch.float32, (m.T + 1, N, m.du)),
                "log_wss": (4, torch.float32, (m.T + 1, N)), "us_star": (5, torch.float32, (m.T + 1, m.du)),
                "vs": (6, torch.float32, (m.T + 1, m.dv))}
        out = {}
        for name, (which, dtype, shape) in spec.items():
            cnt = C.c_int64()
            _lib.call("fbsmi_lg_sweep_view", self.h, which, None, C.byref(cnt), ops._stream())
            if cnt.value == 0:
                out[name] = None
                continue
            buf = torch.empty(cnt.value, dtype=dtype, device=m.device)
            _lib.call("fbsmi_lg_sweep_view", self.h, which, buf.data_ptr(), C.byref(cnt), ops._stream())
            out[name] = self._sq(buf.reshape((Cn,) + shape))
        return out

    def profile(self, enable: bool):
        if self.children:
            raise RuntimeError("profile one group of a grouped batch: sweep.children[0].profile(...)")
        _lib.call("fbsmi_lg_sweep_profile", self.h, int(bool(enable)))

    def kernel_us(self, which: int):
        if self.children:
            return self.children[0].kernel_us(which)
        avg = C.c_double()
        n = C.c_int64()
        _lib.call("fbsmi_lg_sweep_kernel_us", self.h, int(which), C.byref(avg), C.byref(n))
        return avg.value, n.value


class LGFilter:
    """Fused bootstrap_filter (flow='bootstrap', smc.py:9-88) / pmcmc_filter_step (flow='pmcmc',
    smc.py:115-158) for the analytic model: one hipGraph replay per call."""

    _FLOW = {"bootstrap": 0, "pmcmc": 1}
    _RES = {"stratified": 0, "systematic": 1}

    def __init__(self, model: LinearGaussianBridge, nparticles, flow, resampling, store_path):
        self.model, self.n, self.flow, self.store = model, nparticles, flow, store_path
        h = C.c_void_p()
        with torch.cuda.device(model.device):
            _lib.call("fbsmi_lg_filter_create", C.byref(model.struct), nparticles, self._FLOW[flow],
                      self._RES[resampling], int(store_path), 1, C.byref(h))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().fbsmi_lg_filter_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def run(self, key, vs, u0s, use_graph=True):
        """-> (particles (n, du), log-likelihood scalar tensor[, filtering path (T+1, n, du)])."""
        m = self.model
        k = np.asarray(key.detach().cpu() if isinstance(key, torch.Tensor) else key).astype(np.uint32).reshape(1, 2)
        kt = torch.from_numpy(k.view(np.int32).copy()).to(m.device)
        vst = m._t(vs).reshape(m.T + 1, m.dv)
        u0t = m._t(u0s).reshape(self.n, m.du)
        uT = torch.empty((self.n, m.du), dtype=torch.float32, device=m.device)
        ell = torch.empty(1, dtype=torch.float32, device=m.device)
        path = torch.empty((m.T + 1, self.n, m.du), dtype=torch.float32, device=m.device) if self.store else None
        _lib.call("fbsmi_lg_filter_run", self.h, kt.data_ptr(), vst.data_ptr(), u0t.data_ptr(), uT.data_ptr(),
                  ell.data_ptr(), path.data_ptr() if path is not None else None, int(bool(use_graph)), ops._stream())
        return (uT, ell.reshape(())) if path is None else (uT, ell.reshape(()), path)
