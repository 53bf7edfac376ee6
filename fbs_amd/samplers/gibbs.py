"""The forward-backward Gibbs sampler (fbs/samplers/gibbs.py).

``gibbs_kernel`` keeps the reference's signature.  When the closures it receives come from a
``fbs_amd.LinearGaussianBridge`` (analytic score) the whole sweep -- forward noising, T-step
conditional SMC with killing resampling, forced move, fresh reference trajectory -- runs fused on
the device as one hipGraph replay.  Any other closures take the generic tier: a host loop whose
sampler-side operations are HIP kernels.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib, ops
from ..sdes.simulators import doob_bridge_simulator
from .csmc.csmc import csmc_kernel, _forward as _csmc_fwd
from .csmc.resamplings import killing
from .resampling import stratified
from .smc import bootstrap_filter, bootstrap_backward_smoother


def bridge_sampler(key, y0, yT, ts, sde):
    """Sampling Doob's h-transform (gibbs.py:17-20)."""
    return doob_bridge_simulator(key, sde, y0, yT, ts, integration_nsteps=100, replace=True)


def gibbs_init(key, y0, x0_shape, ts, fwd_sampler, sde, unpack, transition_sampler, transition_logpdf,
               likelihood_logpdf, nparticles, method: str = 'smoother', marg_y: bool = True, x0=None, **kwargs):
    """Initialise the Gibbs sampler with a draw from a bootstrap filter/smoother (gibbs.py:23-65)."""
    device = y0.device if isinstance(y0, torch.Tensor) else ops._default_device()
    if x0 is None:
        x0 = torch.zeros(x0_shape, dtype=torch.float32, device=device)
    key_fwd, key_bridge, key_u0, key_bf, key_fwd2, key_bwd = ops.split(key, 6)      # :39
    path_xy = fwd_sampler(key_fwd, x0, y0, **kwargs)
    _, path_y = unpack(path_xy, **kwargs)
    vs = torch.flip(bridge_sampler(key_bridge, path_y[0], path_y[-1], ts, sde), [0]) if marg_y \
        else torch.flip(path_y, [0])                                                # :44

    def init_sampler(*_):  # :46-48: ignores its arguments, always key_u0
        return ops.normal(key_u0, (nparticles,) + tuple(x0_shape), device=device)

    if method == 'filter':
        approx_x0 = bootstrap_filter(transition_sampler, likelihood_logpdf, vs, ts, init_sampler, key_bf, nparticles,
                                     stratified, log=True, return_last=True, **kwargs)[0][0]
        approx_us_star = torch.flip(unpack(fwd_sampler(key_fwd2, approx_x0, y0, **kwargs), **kwargs)[0], [0])
    elif method == 'smoother':
        uss = bootstrap_filter(transition_sampler, likelihood_logpdf, vs, ts, init_sampler, key_bf, nparticles,
                               stratified, log=True, return_last=False, **kwargs)[0]
        approx_x0 = uss[-1, 0]
        approx_us_star = bootstrap_backward_smoother(key_bwd, uss, vs, ts, transition_logpdf, **kwargs)
    elif method == 'debug':
        approx_x0 = bootstrap_filter(transition_sampler, likelihood_logpdf, vs, ts, init_sampler, key_bf, nparticles,
                                     stratified, log=True, return_last=False, **kwargs)[0]
        approx_us_star = None
    else:
        raise ValueError(f"Unknown method {method}")
    return approx_x0, approx_us_star


def _lg_model_of(*closures):
    """The LinearGaussianBridge all closures belong to, or None."""
    models = [getattr(c, "_fbsmi_lg", None) for c in closures]
    if any(m is None for m in models):
        return None
    return models[0] if all(m is models[0] for m in models) else None


def _same_sde(a, b) -> bool:
    """The same SDE object, or one of the same class with the same coefficients (scalars or arrays)."""
    if a is b:
        return True
    if type(a) is not type(b):
        return False
    va, vb = vars(a), vars(b)
    if va.keys() != vb.keys():
        return False
    for k in va:
        x, y = va[k], vb[k]
        if isinstance(x, torch.Tensor) or isinstance(y, torch.Tensor):
            x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
            y = y.detach().cpu().numpy() if isinstance(y, torch.Tensor) else y
        try:
            if not np.array_equal(np.asarray(x), np.asarray(y)):
                return False
        except Exception:
            return False
    return True


def gibbs_kernel(key, x0, y0, us_star, bs_star, ts, fwd_sampler, sde, unpack, nparticles, transition_sampler,
                 transition_logpdf, likelihood_logpdf, marg_y: bool = False, explicit_backward: bool = True,
                 explicit_final: bool = False, **kwargs):
    """Gibbs kernel of the forward-backward conditional sampler (gibbs.py:68-168).

    Returns (x0, us_star, bs_star, acc) like the reference."""
    # the fused engine runs the model's OWN grid, SDE and split: take it only when the caller passed exactly those
    model = _lg_model_of(fwd_sampler, transition_sampler, likelihood_logpdf, unpack)
    # (with marg_y=False the reference never touches `sde` (gibbs.py:130) and its drivers may pass None, experiments/sb/gibbs.py:171)
    if model is not None and not marg_y and not kwargs and (sde is None or _same_sde(sde, model.sde)) and model.same_grid(ts) and \
            (explicit_backward or _lg_model_of(transition_logpdf) is model) and \
            model.fused_sweep_supported(nparticles, explicit_final):
        with torch.cuda.device(model.device):
            return model.gibbs_kernel(key, x0, y0, bs_star, nparticles, explicit_backward, explicit_final)

    key_fwd, key_csmc, key_bridge = ops.split(key, 3)                               # :126
    path_xy = fwd_sampler(key_fwd, x0, y0, **kwargs)                                # :127
    path_x, path_y = unpack(path_xy, **kwargs)
    us = torch.flip(path_x, [0])                                                    # :129
    vs = torch.flip(bridge_sampler(key_bridge, path_y[0], path_y[-1], ts, sde), [0]) if marg_y \
        else torch.flip(path_y, [0])                                                # :130
    ts0 = ts[0]

    if explicit_final:                                                              # :132-138
        def init_sampler(key_, n_samples):
            return ops.normal(key_, (n_samples,) + tuple(us.shape[1:]), device=us.device)

        def init_likelihood_logpdf(v0, u0s, v1, **kw):
            return likelihood_logpdf(v0, u0s, v1, ts0, **kw)
    else:                                                                           # :139-144
        def init_sampler(*_):
            return us[0].unsqueeze(0).expand((nparticles,) + tuple(us.shape[1:])).clone()

        def init_likelihood_logpdf(*_, **__):
            return torch.full((nparticles,), -math.log(nparticles), dtype=torch.float32, device=us.device)

    bs_np = np.asarray(bs_star.detach().cpu() if isinstance(bs_star, torch.Tensor) else bs_star).reshape(-1)
    if explicit_backward:
        k_fwd, k_x0, k_us, k_bs = ops.split(key_csmc, 4)                            # :147
        _, log_ws_T, us_T = _csmc_fwd(k_fwd, us, bs_np, vs, ts, init_sampler, init_likelihood_logpdf,
                                      transition_sampler, likelihood_logpdf, killing, nparticles, False, **kwargs)
        idx, _ = force_move(k_x0, ops.math_map("exp", log_ws_T), int(bs_np[-1]))    # :152
        x0 = us_T[idx.long()]                                                       # :154
        us_star_next = torch.flip(unpack(fwd_sampler(k_us, x0, y0, **kwargs), **kwargs)[0], [0])  # :155
        bs_star_next = ops.randint(k_bs, (us.shape[0],), 0, nparticles, device=us.device)          # :156
    else:
        us_star_next, bs_star_next = csmc_kernel(key_csmc, us, bs_np, vs, ts, init_sampler, init_likelihood_logpdf,
                                                 transition_sampler, transition_logpdf, likelihood_logpdf, killing,
                                                 nparticles, backward=False, **kwargs)
    x0_next = us_star_next[-1]
    bs_old = torch.as_tensor(bs_np.astype(np.int32), device=bs_star_next.device)
    return x0_next, us_star_next, bs_star_next, bs_star_next != bs_old              # :167-168


def force_move(key, weights, k):
    """Forced-move trajectory selection (gibbs.py:171-214) -> (index tensor, alpha tensor)."""
    w = ops._f32c(weights, "weights").reshape(-1)
    out_i = torch.empty(1, dtype=torch.int32, device=w.device)
    out_a = torch.empty(1, dtype=torch.float32, device=w.device)
    k0, k1 = ops._k(key)
    kk = int(k.item()) if isinstance(k, torch.Tensor) else int(k)
    _lib.call("fbsmi_force_move", k0, k1, w.data_ptr(), kk, w.numel(), out_i.data_ptr(), out_a.data_ptr(),
              ops._ws(w.numel(), w.device).data_ptr(), ops._stream())
    return out_i.reshape(()), out_a.reshape(())
