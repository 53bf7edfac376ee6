"""Conditional score-based generative sampling for image restoration (Song et al., 2021): the `csgm` baseline of the
reference's image tables.

Counterpart of experiments/imgs/inpainting_csgm.py:88-124 and supr_csgm.py (they differ by the mask only): ONE trajectory of
the reverse SDE of the unobserved pixels, the observed pixels replaced at every step by a fresh noising of y0 to the current
time.  No particles, no resampling: per step one draw of the observed part, one network evaluation and one Euler-Maruyama
update whose noise is the step's slice of normal(key_scan, (nsteps, *x_shape)), drawn inside the kernel (fbsmi_em_update).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib, ops
from .sdes import make_linear_sde


def make_image_csgm(score_fn, ds, sde, ts):
    """score_fn(uv (1, w, h, c), t) -> score of the same shape.  -> conditional_sampler(key, y0, mask_) -> x0 (p, c)."""
    ts = np.asarray(ts, np.float64)
    T = float(ts[-1])
    nsteps = ts.size - 1
    dt = T / nsteps                                                                   # inpainting_csgm.py:49
    discretise = make_linear_sde(sde)[0]
    x_shape = tuple(ds.unobs_shape)

    def reverse_drift(u, t, mask_, key_, y0):                                         # :88-92
        s_ = T - float(t)
        F, Q = (float(x) for x in discretise(s_, float(ts[0])))
        v_hat = F * y0 + math.sqrt(Q) * ops.normal(key_, tuple(y0.shape), device=y0.device)
        uv = ds.concat(u, v_hat, mask_)
        score_u = ds.unpack(score_fn(uv.unsqueeze(0), s_).reshape(uv.shape).float(), mask_)[0]
        return -sde.drift(u, s_) + float(sde.dispersion(s_)) ** 2 * score_u

    def euler_maruyama(key_, u0, mask_, y0):                                          # :103-113
        key_scan, key_est = ops.split(key_)
        key_ests = ops.split(key_est, nsteps)
        k0, k1 = (int(x) for x in key_scan)
        numel = int(u0.numel())
        u = u0.contiguous()
        for k in range(nsteps):
            t = float(ts[k])
            f = reverse_drift(u, t, mask_, key_ests[k], y0).to(torch.float32).contiguous()
            c = float(np.float32(float(sde.dispersion(T - t)) * math.sqrt(dt)))
            out = torch.empty_like(u)
            _lib.call("fbsmi_em_update", u.data_ptr(), f.data_ptr(), float(np.float32(dt)), c, k0, k1, nsteps * numel,
                      k * numel, numel, out.data_ptr(), ops._stream())
            u = out
        return u

    def conditional_sampler(key_, y0, mask_):                                         # :116-121
        key_init, key_sde = ops.split(key_, 2)
        u0 = ops.normal(key_init, x_shape, device=y0.device)                          # cond_ref_sampler :99-100
        return euler_maruyama(key_sde, u0, mask_, y0)

    return conditional_sampler
