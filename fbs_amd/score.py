"""Model closures around a score / drift network for the image tasks (experiments/imgs/inpainting.py:98-161,
supr.py identical up to the mask; experiments/sb_imgs/supr.py:80-141 for the Schrodinger-bridge model).

Per SMC step the reference evaluates the network twice on the same input (inside ``transition_sampler``
and inside ``likelihood_logpdf``: csmc.py:142,145) and spreads concat / unpack / drift / Euler-Maruyama /
norm.logpdf / sum over a dozen array passes.  Here a step is

    fbsmi_em_concat  (ancestor gather + concat -> the network input, written once)
    the network      (PyTorch-ROCm, in chunks of `chunk` particles)
    fbsmi_em_finish  (unpack + drift + proposal with in-kernel normal draw + pin + row-summed log-density)

``ScoreBridge.fused_step`` is that sequence; ``fbs_amd.samplers`` calls it when it is handed this bridge's
closures.  The three closures themselves keep the reference's signatures and run on the same kernels
(the network output of a given (us_prev, v_prev, t_prev, mask) is computed once and shared).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib, ops

_NET_DT = {torch.float32: 0, torch.bfloat16: 1}


class EMMask:
    """Device tables of a mask for the fused kernels (include/fbsmi.h, fbsmi_em_mask)."""

    def __init__(self, mask, channels: int, device):
        c = int(channels)
        unobs = mask.unobs_inds_ravelled.detach().cpu().numpy().astype(np.int64)
        obs = mask.obs_inds_ravelled.detach().cpu().numpy().astype(np.int64)
        u_off = (unobs[:, None] * c + np.arange(c)[None, :]).reshape(-1)
        v_off = (obs[:, None] * c + np.arange(c)[None, :]).reshape(-1)
        D = u_off.size + v_off.size
        role = np.full(D, np.iinfo(np.int64).min, np.int64)
        role[u_off] = np.arange(u_off.size)
        role[v_off] = ~np.arange(v_off.size)
        if (role == np.iinfo(np.int64).min).any():
            raise ValueError("the mask does not cover every pixel exactly once")
        self.mask = mask                                   # keeps the key object alive
        self.du, self.dv, self.D = int(u_off.size), int(v_off.size), int(D)
        dev = lambda a: torch.from_numpy(a.astype(np.int32)).to(device)
        self.u_off, self.v_off, self.role = dev(u_off), dev(v_off), dev(role)
        self.struct = _lib.EMMaskStruct(self.du, self.dv, self.u_off.data_ptr(), self.v_off.data_ptr(),
                                        self.role.data_ptr())

    @property
    def ref(self):
        return C.byref(self.struct)


class ScoreBridge:
    def __init__(self, score_fn, dataset, sde, ts, chunk: int = 1024, mode: str = "score",
                 net_input_dtype=torch.float32, fwd_drift_fn=None):
        """score_fn(x (B, w, h, c), t float) -> (B, w, h, c), float32 or bfloat16.

        mode 'score': score_fn is the trained score at forward time t and the reverse drift is
        -sde.drift + dispersion^2 * score (inpainting.py:102-103); mode 'drift': score_fn is the learned
        backward drift itself (sb_imgs/supr.py:84-85) and ``fwd_drift_fn(x, t)`` the learned forward drift
        that ``fwd_sampler`` integrates (supr.py:132-137).  net_input_dtype: dtype the network input is
        written in (torch.bfloat16 saves the cast pass of an autocast network)."""
        if mode not in ("score", "drift"):
            raise ValueError(f"unknown mode {mode}")
        self.score_fn, self.dataset, self.sde = score_fn, dataset, sde
        self.mode, self.fwd_drift_fn = mode, fwd_drift_fn
        self.ts = np.asarray(ts, np.float64)
        self.T = float(self.ts[-1])
        self.nsteps = self.ts.size - 1
        self.dt = self.T / self.nsteps                       # inpainting.py:58-59
        self.chunk = int(chunk)
        self.net_input_dtype = net_input_dtype
        self._cache = None
        self._masks = {}
        self._buf = {}
        self.profile = None                                  # set to a dict to collect torch events per phase
        self.capture = None                                  # set to a dict: operands and results of the LAST fused step

    # -- plumbing -----------------------------------------------------------------------------------
    def _em_mask(self, mask_) -> EMMask:
        m = self._masks.get(id(mask_))
        if m is None or m.mask is not mask_:
            m = EMMask(mask_, self.dataset.image_shape[2], self.dataset.device)
            self._masks[id(mask_)] = m
        return m

    def _buffer(self, name, shape, dtype, device):
        b = self._buf.get(name)
        if b is None or b.shape != tuple(shape) or b.dtype != dtype or b.device != device:
            b = torch.empty(tuple(shape), dtype=dtype, device=device)
            self._buf[name] = b
        return b

    def _coef(self, t_prev):
        """(mode, cx, cs, sd) of the step that leaves reverse time t_prev."""
        tf = self.T - float(t_prev)
        sd = math.sqrt(self.dt) * float(self.sde.dispersion(tf))
        if self.mode == "drift":
            return 1, 0.0, 1.0, sd
        return 0, -float(self.sde.drift(1.0, tf)), float(self.sde.dispersion(tf)) ** 2, sd

    def _mark(self, name):
        if self.profile is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.profile.setdefault(name, []).append(e)
        return e

    @torch.no_grad()
    def _network(self, us, A, v_prev, t_prev, em: EMMask) -> torch.Tensor:
        """nn(concat(us[A], v_prev), T - t) for every row -> (n, D), the network's own output dtype."""
        us = ops._f32c(us, "us")
        n = int(A.numel()) if A is not None else int(us.shape[0])
        w, h, c = self.dataset.image_shape
        img = self._buffer("img", (n, w, h, c), self.net_input_dtype, us.device)
        vp = ops._f32c(v_prev, "v_prev")
        self._mark("concat0")
        _lib.call("fbsmi_em_concat", em.ref, us.data_ptr(), A.data_ptr() if A is not None else None, vp.data_ptr(), n,
                  _NET_DT[self.net_input_dtype], img.data_ptr(), ops._stream())
        self._mark("concat1")
        tf = self.T - float(t_prev)
        out = None
        # Fixed chunks of `chunk` rows plus a tail (explicit_final's row N + 1 is a call of its own when chunk divides N):
        # one call of N + 1 rows is 8 % faster per step once MIOpen has built kernels for that batch size, but on a fresh
        # machine every unusual batch size costs tens of seconds of kernel builds (measured: +50 s / +120 s for a sweep of
        # configs 3 / 5) -- callers that run many sweeps pass chunk >= N + 1.
        per = self.chunk
        for s in range(0, n, per):
            y = self.score_fn(img[s:s + per], tf)
            if out is None:
                if y.dtype not in _NET_DT:
                    y = y.float()
                out = self._buffer("net", (n, em.D), y.dtype, us.device)
            out[s:s + per] = y.reshape(-1, em.D)
        self._mark("net1")
        return out

    def _finish(self, em, us, A, net, t_prev, v, v_prev, key_, row_slice, pin, want_us, want_lw, net_A=None):
        mode, cx, cs, sd = self._coef(t_prev)
        n = int(net_A.numel() if net_A is not None else net.shape[0])
        row0, cnt, tot = (0, n, n) if row_slice is None else (int(row_slice[0]), int(row_slice[1]), int(row_slice[2]))
        if cnt != n:
            raise ValueError(f"row_slice says {cnt} local rows, the ensemble shard has {n}")
        us_new = torch.empty((n, em.du), dtype=torch.float32, device=net.device) if want_us else None
        lw = torch.empty(n, dtype=torch.float32, device=net.device) if want_lw else None
        k0, k1 = ops._k(key_) if want_us else (0, 0)
        pin_row, pin_val = (-1, None) if pin is None else (int(pin[0]), ops._f32c(pin[1], "pin value"))
        vv = ops._f32c(v, "v") if want_lw else None
        vp = ops._f32c(v_prev, "v_prev") if want_lw else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        self._mark("finish0")
        _lib.call("fbsmi_em_finish", em.ref, ptr(us), ptr(A), net.data_ptr(), ptr(net_A), _NET_DT[net.dtype], mode, cx, cs,
                  self.dt, sd, ptr(vv), ptr(vp), k0, k1, tot, row0, n, pin_row, ptr(pin_val), ptr(us_new), ptr(lw),
                  ops._stream())
        self._mark("finish1")
        return us_new, lw

    # -- the fused step -------------------------------------------------------------------------------
    def fused_step(self, us, A, v, v_prev, t_prev, key_, mask_, pin=None, row_slice=None, want_lw=True):
        """One SMC step of csmc.forward_pass (csmc.py:140-145) / smc.py:63-65 around one network evaluation:
        us_prev = us[A] (A None: identity), proposal, pin = (local row, value) or None, log-weights of
        the gathered particles.  -> (us_new (n, p, c), lw (n,))."""
        em = self._em_mask(mask_)
        self._cache = None                                   # the shared network buffer is about to be rewritten
        us2 = ops._f32c(us, "us").reshape(us.shape[0], -1)
        A32 = A.to(torch.int32).contiguous() if A is not None else None
        net = self._network(us2, A32, v_prev, t_prev, em)
        us_new, lw = self._finish(em, us2, A32, net, t_prev, v, v_prev, key_, row_slice, pin, True, want_lw)
        if self.capture is not None:                         # parity tests replay this step through the oracle
            self.capture.update(us=us2, A=A32, net=net, img=self._buf["img"], v=v, v_prev=v_prev, t_prev=float(t_prev),
                                key=np.asarray(key_, np.uint32).copy(), pin=pin, row_slice=row_slice, us_new=us_new,
                                lw=lw, coef=self._coef(t_prev), em=em)
        return us_new.reshape((us_new.shape[0],) + tuple(self.dataset.unobs_shape)), lw

    def fused_weight_then_propose(self, us, v, v_prev, t_prev, key_, mask_, resample):
        """One step of pmcmc_filter_step (fbs/samplers/smc.py:144-150): weight the particles, resample,
        propose from the resampled ones.  The reference evaluates the network on `us` for the weights and
        again on `us[inds]` for the proposal; the second input is a row gather of the first, so the network
        runs once and the proposal reads its output rows through `inds`.
        resample(log_ws (n,)) -> inds (n,) int32.  -> (us_new, log_ws (unnormalised), inds)."""
        em = self._em_mask(mask_)
        self._cache = None
        us2 = ops._f32c(us, "us").reshape(us.shape[0], -1)
        net = self._network(us2, None, v_prev, t_prev, em)
        _, lw = self._finish(em, None, None, net, t_prev, v, v_prev, None, None, None, False, True)
        inds = resample(lw).to(torch.int32).contiguous()
        us_new, _ = self._finish(em, us2, inds, net, t_prev, None, None, key_, None, None, True, False, net_A=inds)
        return us_new.reshape((us_new.shape[0],) + tuple(self.dataset.unobs_shape)), lw, inds

    # -- network output shared by the closures of one step --------------------------------------------
    def _net_of(self, us_prev, v_prev, t_prev, mask_):
        c = self._cache
        hit = (c is not None and c["us"] is us_prev and c["usv"] == us_prev._version and c["v"] is v_prev
               and c["vv"] == v_prev._version and c["t"] == float(t_prev) and c["mask"] is mask_)
        if not hit:
            em = self._em_mask(mask_)
            net = self._network(us_prev.reshape(us_prev.shape[0], -1), None, v_prev, t_prev, em)
            # the entry owns references to its key tensors, so their storage cannot be recycled under it
            self._cache = c = {"us": us_prev, "usv": us_prev._version, "v": v_prev, "vv": v_prev._version,
                               "t": float(t_prev), "mask": mask_, "net": net, "em": em}
        return c["em"], c["net"]

    def reverse_drift(self, uv: torch.Tensor, t: float) -> torch.Tensor:      # inpainting.py:102-103
        """Reverse drift of joint images uv (B, w, h, c) at reverse time t (diagnostics; the samplers use
        the fused kernels)."""
        tf = self.T - float(t)
        out = torch.empty(uv.shape, dtype=torch.float32, device=uv.device)
        with torch.no_grad():
            for s in range(0, uv.shape[0], self.chunk):
                x = uv[s:s + self.chunk]
                y = self.score_fn(x.to(self.net_input_dtype), tf).float().reshape(x.shape)
                out[s:s + self.chunk] = y if self.mode == "drift" else \
                    -float(self.sde.drift(1.0, tf)) * x + float(self.sde.dispersion(tf)) ** 2 * y
        return out

    def reverse_dispersion(self, t):                                           # :118-119
        return float(self.sde.dispersion(self.T - float(t)))

    # -- closures (signatures of the reference, mask threaded through **kwargs as `mask_`) ----------
    def unpack(self, xy, mask_):
        return self.dataset.unpack(xy, mask_)

    def transition_sampler(self, us_prev, v_prev, t_prev, key_, mask_, row_slice=None):   # :122-128
        em, net = self._net_of(us_prev, v_prev, t_prev, mask_)
        us2 = ops._f32c(us_prev, "us_prev").reshape(us_prev.shape[0], -1)
        us_new, _ = self._finish(em, us2, None, net, t_prev, None, None, key_, row_slice, None, True, False)
        return us_new.reshape(us_prev.shape)

    def transition_logpdf(self, u, u_prev, v_prev, t_prev, mask_):                        # :131-138
        em, net = self._net_of(u_prev, v_prev, t_prev, mask_)
        mode, cx, cs, sd = self._coef(t_prev)
        us2 = ops._f32c(u_prev, "u_prev").reshape(u_prev.shape[0], -1)
        uu = ops._f32c(u, "u")
        lw = torch.empty(us2.shape[0], dtype=torch.float32, device=us2.device)
        _lib.call("fbsmi_em_transition_logpdf", em.ref, us2.data_ptr(), net.data_ptr(), _NET_DT[net.dtype], mode, cx,
                  cs, self.dt, sd, uu.data_ptr(), us2.shape[0], lw.data_ptr(), ops._stream())
        return lw

    def likelihood_logpdf(self, v, u_prev, v_prev, t_prev, mask_):                        # :141-147
        em, net = self._net_of(u_prev, v_prev, t_prev, mask_)
        _, lw = self._finish(em, None, None, net, t_prev, v, v_prev, None, None, None, False, True)
        return lw

    def fwd_sampler(self, key_, x0_, y0_, mask_):                                         # :150-152
        xy0 = self.dataset.concat(x0_.unsqueeze(0), y0_, mask_)[0]
        if self.mode == "drift":                                                          # sb_imgs/supr.py:132-137
            from .sdes.simulators import euler_maruyama
            return euler_maruyama(key_, xy0, self.ts, self.fwd_drift_fn, self.sde.dispersion, integration_nsteps=1,
                                  return_path=True)
        from .sdes import make_linear_sde
        return make_linear_sde(self.sde)[2](key_, xy0, self.ts)

    def fwd_ys_sampler(self, key_, y0_):                                                  # :155-157
        from .sdes import make_linear_sde
        return make_linear_sde(self.sde)[2](key_, y0_, self.ts)

    def ref_sampler(self, key_, _, n, row_slice=None):                                    # :160-161
        shape = (n,) + tuple(self.dataset.unobs_shape)
        return ops.normal(key_, shape, device=self.dataset.device, rows=None if row_slice is None else row_slice[:2])


def bridge_of(*closures):
    """The ScoreBridge whose OWN transition_sampler / likelihood_logpdf (in that order; a third closure, if given, its
    transition_logpdf) the given closures are, or None.  Callers replace the pair by ``fused_step``, which has exactly those two
    methods' semantics -- so a subclass override, another method of the bridge (e.g. transition_logpdf passed as the weight
    function) or a swapped pair must not be taken for them."""
    owners = [getattr(c, "__self__", None) for c in closures]
    if not owners or not isinstance(owners[0], ScoreBridge) or not all(o is owners[0] for o in owners):
        return None
    want = (ScoreBridge.transition_sampler, ScoreBridge.likelihood_logpdf, ScoreBridge.transition_logpdf)
    if len(closures) > len(want):
        return None
    for c, w in zip(closures, want):
        if getattr(c, "__func__", None) is not w:
            return None
    return owners[0]
