"""Model closures around a score network for the image tasks (experiments/imgs/inpainting.py:98-161,
supr.py identical up to the mask): reverse drift of the joint (U, V) image through the network,
Euler-Maruyama proposal of the unobserved pixels, Gaussian log-weight of the observed increment.

The reference evaluates the network twice per SMC step on the same input (once inside
``transition_sampler``, once inside ``likelihood_logpdf``: csmc.py:142,145).  Here the drift of a
given (us_prev, v_prev, t_prev) is computed once and cached for the second closure, and the N
particles go through the network in chunks so that N = 16 384 images fit the activations.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


class ScoreBridge:
    def __init__(self, score_fn, dataset, sde, ts, chunk: int = 1024):
        """score_fn(x (B, w, h, c), t float) -> (B, w, h, c): the trained score at forward time t."""
        self.score_fn, self.dataset, self.sde = score_fn, dataset, sde
        self.ts = np.asarray(ts, np.float64)
        self.T = float(self.ts[-1])
        self.nsteps = self.ts.size - 1
        self.dt = self.T / self.nsteps                       # inpainting.py:58-59
        self.chunk = int(chunk)
        self._cache = {}

    # -- drift of the joint image at reverse time t -------------------------------------------------
    @torch.no_grad()
    def reverse_drift(self, uv: torch.Tensor, t: float) -> torch.Tensor:      # inpainting.py:102-103
        tf = self.T - float(t)
        a = float(self.sde.drift(1.0, tf))
        b2 = float(self.sde.dispersion(tf)) ** 2
        out = torch.empty_like(uv)
        for s in range(0, uv.shape[0], self.chunk):
            x = uv[s:s + self.chunk]
            sc = self.score_fn(x, tf)
            out[s:s + self.chunk] = -a * x + b2 * sc.reshape(x.shape)
        return out

    def _drift_uv(self, us_prev, v_prev, t_prev, mask_):
        key = (us_prev.data_ptr(), us_prev._version, v_prev.data_ptr(), float(t_prev), tuple(us_prev.shape))
        hit = self._cache.get("k") == key
        if not hit:
            n = us_prev.shape[0]
            img = self.dataset.concat(us_prev, v_prev, mask_)                  # (n, w, h, c)
            rdu, rdv = self.dataset.unpack(self.reverse_drift(img, t_prev), mask_)
            self._cache = {"k": key, "rdu": rdu, "rdv": rdv}
        return self._cache["rdu"], self._cache["rdv"]

    def reverse_dispersion(self, t):                                           # :118-119
        return float(self.sde.dispersion(self.T - float(t)))

    # -- closures (signatures of the reference, mask threaded through **kwargs as `mask_`) ----------
    def unpack(self, xy, mask_):
        return self.dataset.unpack(xy, mask_)

    def transition_sampler(self, us_prev, v_prev, t_prev, key_, mask_, row_slice=None):   # :122-128
        rdu, _ = self._drift_uv(us_prev, v_prev, t_prev, mask_)
        if row_slice is None:
            z = ops.normal(key_, tuple(us_prev.shape), device=us_prev.device)
        else:
            off, cnt, tot = row_slice
            z = ops.normal(key_, (tot,) + tuple(us_prev.shape[1:]), device=us_prev.device, rows=(off, cnt))
        return us_prev + rdu * self.dt + (math.sqrt(self.dt) * self.reverse_dispersion(t_prev)) * z

    @staticmethod
    def _norm_logpdf_sum(x, loc, scale):
        # jax.scipy.stats.norm.logpdf summed over the pixel axes
        var = scale * scale
        lp = (math.log(2 * math.pi * var) + (x - loc) ** 2 / var) / -2.0
        return lp.reshape(lp.shape[0], -1).sum(dim=1)

    def transition_logpdf(self, u, u_prev, v_prev, t_prev, mask_):                        # :131-138
        rdu, _ = self._drift_uv(u_prev, v_prev, t_prev, mask_)
        return self._norm_logpdf_sum(u.unsqueeze(0), u_prev + rdu * self.dt,
                                     math.sqrt(self.dt) * self.reverse_dispersion(t_prev))

    def likelihood_logpdf(self, v, u_prev, v_prev, t_prev, mask_):                        # :141-147
        _, rdv = self._drift_uv(u_prev, v_prev, t_prev, mask_)
        cond_m = v_prev.unsqueeze(0) + rdv * self.dt
        return self._norm_logpdf_sum(v.unsqueeze(0), cond_m, math.sqrt(self.dt) * self.reverse_dispersion(t_prev))

    def fwd_sampler(self, key_, x0_, y0_, mask_):                                         # :150-152
        from .sdes import make_linear_sde
        xy0 = self.dataset.concat(x0_.unsqueeze(0), y0_, mask_)[0]
        return make_linear_sde(self.sde)[2](key_, xy0, self.ts)

    def fwd_ys_sampler(self, key_, y0_):                                                  # :155-157
        from .sdes import make_linear_sde
        return make_linear_sde(self.sde)[2](key_, y0_, self.ts)

    def ref_sampler(self, key_, _, n):                                                    # :160-161
        return ops.normal(key_, (n,) + tuple(self.dataset.unobs_shape), device=self.dataset.device)
