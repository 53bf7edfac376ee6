"""Twisted-SMC closures for image restoration with a score network.

Counterpart of experiments/imgs/inpainting_twisted.py:97-154 (the Wu et al., 2023 baseline of the reference's tables): the
particles are whole images (n, w, h, c); the twisting function is the Gaussian likelihood of the observed pixels under the
one-step denoising estimate, and the proposal's drift adds its gradient, which goes THROUGH the score network -- by
torch.autograd here, by jax.grad in the reference.  `fbs_amd.samplers.smc.twisted_smc` consumes the closures unchanged;
resampling, gathers, normalisation and the noise draws are libfbsmi kernels, the network stays a PyTorch-ROCm module.

The reference evaluates the score up to four times per step on the same batch (transition_logpdf, twisting_logpdf, the
proposal's sampler and its log-density; XLA's CSE merges them under jit).  Here the results are remembered per
(tensor object, time): one forward evaluation and one forward + backward evaluation per step.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .sdes import make_linear_sde


def norm_logpdf_rows(x, loc, scale: float):
    """sum over everything but the leading axis of jax.scipy.stats.norm.logpdf(x, loc, scale)."""
    z = (x - loc) / scale
    lp = -0.5 * z * z - math.log(scale) - 0.5 * math.log(2.0 * math.pi)
    return lp.reshape(lp.shape[0], -1).sum(dim=1)


def make_image_twisted(score_fn, ds, sde, ts, nparticles: int, data_variance: float = 0.06):
    """score_fn(uv (n, w, h, c), t) -> score, differentiable with respect to uv (a torch module call WITHOUT no_grad).

    -> namespace(init_sampler, transition_logpdf, twisting_logpdf, twisting_prop_sampler, twisting_prop_logpdf,
                 conditional_sampler), each with the reference's signature; `mask_` is threaded as a keyword."""
    ts = np.asarray(ts, np.float64)
    T = float(ts[-1])
    dt = T / (ts.size - 1)                                                            # inpainting_twisted.py:51
    xy_shape = tuple(ds.image_shape)
    discretise = make_linear_sde(sde)[0]
    memo = {}

    def cached(tag, x, t, fn):
        k = (tag, id(x), round(float(t), 12))
        hit = memo.get(k)
        if hit is not None and hit[0] is x:
            return hit[1]
        if len(memo) > 8:
            memo.clear()
        out = fn()
        memo[k] = (x, out)
        return out

    def reverse_drift(uv, t):                                                         # :97-98
        s_ = T - float(t)
        return -sde.drift(uv, s_) + float(sde.dispersion(s_)) ** 2 * score_fn(uv, s_)

    def reverse_dispersion(t):                                                        # :106-107
        return float(sde.dispersion(T - float(t)))

    def _obs_layout(y, mask_):
        """(y scattered into image layout, 1 on observed pixels / 0 elsewhere), both (1, w, h, c): the observed part of an
        image then is an elementwise product -- what the differentiable section uses instead of a gather, so that its backward
        pass is elementwise too."""
        k = (id(y), id(mask_))
        hit = memo.get(("obs",) + k)
        if hit is not None and hit[0] is y:
            return hit[1]
        w, h, c = xy_shape
        yfull = torch.zeros((w * h, c), dtype=torch.float32, device=y.device)
        yfull.index_copy_(0, mask_.obs_inds_ravelled, y.reshape(-1, c).float())
        m = torch.zeros((w * h, 1), dtype=torch.float32, device=y.device)
        m.index_fill_(0, mask_.obs_inds_ravelled, 1.0)
        out = (yfull.reshape(1, w, h, c), m.reshape(1, w, h, 1))
        memo[("obs",) + k] = (y, out)
        return out

    def _twist(y, uv, t, mask_):                                                      # :122-126
        den = uv + reverse_drift(uv, t) * dt
        F, Q = discretise(T - float(t), float(ts[0]))
        scale = math.sqrt(float(F) ** 2 * data_variance + float(Q))
        yfull, m = _obs_layout(y, mask_)
        z = (yfull - den) / scale                     # observed pixels: norm.logpdf(y, obs_part, scale); the others: 0
        lp = (-0.5 * z * z - math.log(scale) - 0.5 * math.log(2.0 * math.pi)) * m
        return lp.reshape(lp.shape[0], -1).sum(dim=1)

    def twisting_logpdf(y, uvs, t, mask_=None):                                       # :129-131
        def run():
            with torch.no_grad():
                return _twist(y, uvs, t, mask_)
        return cached("tw", uvs, t, run)

    def reverse_cond_drift(uvs, t, y, mask_):                                         # :101-103
        def run():
            # the library convolutions' backward kernels are kept out of this section (torch's own are used): a memory access
            # fault was observed in the backward pass of the small network of the tests with them (round 3; the forward pass
            # and the inference path are unaffected).  This baseline is not on the timed path.
            lib_convs = torch.backends.cudnn.enabled
            torch.backends.cudnn.enabled = False
            try:
                with torch.enable_grad():
                    x = uvs.detach().requires_grad_(True)
                    rd = reverse_drift(x, t)
                    F, Q = discretise(T - float(t), float(ts[0]))
                    scale = math.sqrt(float(F) ** 2 * data_variance + float(Q))
                    yfull, m = _obs_layout(y, mask_)
                    z = (yfull - (x + rd * dt)) / scale
                    lp = ((-0.5 * z * z - math.log(scale) - 0.5 * math.log(2.0 * math.pi)) * m).sum()
                    grad = torch.autograd.grad(lp, x)[0]
            finally:
                torch.backends.cudnn.enabled = lib_convs
            return (rd.detach() + reverse_dispersion(t) ** 2 * grad).float()
        return cached("cd", uvs, t, run)

    def transition_logpdf(u, u_prev, t_prev):                                         # :110-115
        def run():
            with torch.no_grad():
                return reverse_drift(u_prev, t_prev).float()
        rd = cached("rd", u_prev, t_prev, run)
        return norm_logpdf_rows(u, u_prev + rd * dt, math.sqrt(dt) * reverse_dispersion(t_prev))

    def init_sampler(key_, n_):                                                       # :118-119
        return ops.normal(key_, (n_,) + xy_shape, device=ds.device)

    def twisting_prop_sampler(key_, uvs, t, y, mask_=None):                           # :134-137
        m_ = uvs + reverse_cond_drift(uvs, t, y, mask_) * dt
        return m_ + math.sqrt(dt) * reverse_dispersion(t) * ops.normal(key_, (nparticles,) + xy_shape, device=ds.device)

    def twisting_prop_logpdf(u, u_prev, t, y, mask_=None):                            # :140-145
        m_ = u_prev + reverse_cond_drift(u_prev, t, y, mask_) * dt
        return norm_logpdf_rows(u, m_, math.sqrt(dt) * reverse_dispersion(t))

    def conditional_sampler(key_, y, resampling, **kwargs):                           # :148-156
        from .samplers.smc import twisted_smc
        key_filter, key_select = ops.split(key_)
        uvs, log_ws = twisted_smc(key_filter, y, ts, init_sampler, transition_logpdf, twisting_logpdf, twisting_prop_sampler,
                                  twisting_prop_logpdf, resampling=resampling, nparticles=nparticles, **kwargs)
        return ops.choice(key_select, uvs, p=ops.math_map("exp", log_ws), axis=0)

    return SimpleNamespace(init_sampler=init_sampler, transition_logpdf=transition_logpdf, twisting_logpdf=twisting_logpdf,
                           twisting_prop_sampler=twisting_prop_sampler, twisting_prop_logpdf=twisting_prop_logpdf,
                           reverse_cond_drift=reverse_cond_drift, reverse_drift=reverse_drift,
                           conditional_sampler=conditional_sampler, dt=dt, T=T)
