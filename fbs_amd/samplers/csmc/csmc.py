"""Conditional SMC kernel (fbs/samplers/csmc/csmc.py) on torch tensors + libfbsmi primitives.

The user closures (``init_sampler``, ``transition_sampler``, ``likelihood_logpdf`` ...) are Python
callables on GPU tensors, exactly as in the reference, so the time loop is a host loop; every
sampler-owned operation inside it -- conditional resampling, ancestor gather, reference pinning,
logsumexp normalisation, the categorical draws and the ancestor back-trace -- is a HIP kernel.
For the linear-Gaussian model the whole loop is fused on the device instead (see
``fbs_amd.linear_gaussian`` and ``fbs_amd.samplers.gibbs_kernel``).
"""
from __future__ import annotations

import numpy as np
import torch

from ... import ops
from ...score import bridge_of


def _bs_list(bs_star):
    if isinstance(bs_star, torch.Tensor):
        return [int(b) for b in bs_star.detach().cpu().tolist()]
    return [int(b) for b in np.asarray(bs_star).reshape(-1).tolist()]


def csmc_kernel(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                transition_logpdf, measurement_cond_logpdf, cond_resampling, nsamples, backward: bool = False,
                **kwargs):
    """Generic cSMC kernel (csmc.py:14-77) -> (xs_star (K+1, ...), bs_star (K+1,))."""
    key_fwd, key_bwd = ops.split(key, 2)
    As, log_ws, xss = forward_pass(key_fwd, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf,
                                   transition_sampler, measurement_cond_logpdf, cond_resampling, nsamples, **kwargs)
    if backward:
        return backward_sampling_pass(key_bwd, transition_logpdf, vs, ts, xss, log_ws, **kwargs)
    return backward_scanning_pass(key_bwd, As, xss, log_ws[-1])


def _forward(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
             likelihood_logpdf, cond_resampling, nsamples, store: bool, **kwargs):
    nsteps = us_star.shape[0] - 1
    bs = _bs_list(bs_star)
    key_init, key_scan = ops.split(key, 2)                                        # csmc.py:150
    us = init_sampler(key_init, nsamples + 1)                                      # :151
    us = ops.set_row(us, bs[0], us_star[0])                                        # :152
    log_ws = ops.normalise(init_likelihood_logpdf(vs[0], us, vs[1], **kwargs), log_space=True)  # :154-155
    keys = ops.split(key_scan, nsteps)                                             # :157
    As, log_wss, uss = [], [log_ws], [us]
    # closures of one fbs_amd.score.ScoreBridge: gather, proposal, pin and weights of a step are two
    # kernels around one network evaluation (fbsmi_em_concat / fbsmi_em_finish)
    sb = bridge_of(transition_sampler, likelihood_logpdf) if set(kwargs) == {"mask_"} else None
    for k in range(nsteps):                                                        # scan_body :132-148
        key_resampling, key_transition = ops.split(keys[k], 2)
        v, v_prev, t_prev = vs[k + 1], vs[k], ts[k]
        A = cond_resampling(key_resampling, ops.math_map("exp", log_ws), bs[k], bs[k + 1], True)  # :139
        if sb is not None:                                                         # :140-145 fused
            us, lw = sb.fused_step(us, A, v, v_prev, t_prev, key_transition, kwargs["mask_"],
                                   pin=(bs[k + 1], us_star[k + 1]))
            log_ws = ops.normalise(lw, log_space=True)                             # :146
        else:
            us_prev = ops.take_rows(us, A)                                         # :140
            us = transition_sampler(us_prev, v_prev, t_prev, key_transition, **kwargs)  # :142
            us = ops.set_row(us, bs[k + 1], us_star[k + 1])                        # :143
            log_ws = ops.normalise(likelihood_logpdf(v, us_prev, v_prev, t_prev, **kwargs), log_space=True)  # :145-146
        if store:
            As.append(A)
            log_wss.append(log_ws)
            uss.append(us)
    if store:
        return torch.stack(As, 0), torch.stack(log_wss, 0), torch.stack(uss, 0)
    return None, log_ws, us


def forward_pass(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                 likelihood_logpdf, cond_resampling, nsamples, **kwargs):
    """Forward pass of the cSMC kernel (csmc.py:80-164) -> (As (K,n), log_wss (K+1,n), uss (K+1,n,...))."""
    return _forward(key, us_star, bs_star, vs, ts, init_sampler, init_likelihood_logpdf, transition_sampler,
                    likelihood_logpdf, cond_resampling, nsamples, True, **kwargs)


def backward_sampling_pass(key, transition_logpdf, vs, ts, uss, log_ws, *args, **kwargs):
    """Backward sampling pass (csmc.py:167-227)."""
    K_plus_one = uss.shape[0]
    keys = ops.split(key, K_plus_one)                                              # :194
    W_T = normalise(log_ws[-1])                                                    # :200
    B_T = barker_move(keys[-1], W_T)                                               # :201
    x_t = uss[-1][B_T.long()]
    xs, Bs = [x_t], [B_T.reshape(())]
    # inps = keys[:-1], uss[-2::-1], log_ws[-2::-1], vs[-2::-1], ts[-2::-1]       :214
    for s in range(K_plus_one - 1):
        t = K_plus_one - 2 - s
        Gamma_log_w = transition_logpdf(x_t, uss[t], vs[t], ts[t], *args, **kwargs)  # :205
        Gamma_log_w = Gamma_log_w - torch.max(Gamma_log_w)                         # :206
        w = normalise(Gamma_log_w + log_ws[t])                                     # :207-208
        B = ops.categorical(keys[s], w)                                            # :209
        x_t = uss[t][B.long()]
        xs.append(x_t)
        Bs.append(B)
    return torch.stack(xs[::-1], 0), torch.stack(Bs[::-1], 0).to(torch.int32)


def backward_scanning_pass(key, As, xss, log_w_T):
    """Backward scanning pass (csmc.py:230-270): B_T ~ Cat(w_T), B_{k-1} = A_k[B_k]."""
    B_T = barker_move(key, normalise(log_w_T))                                     # :257
    Bs = ops.backtrace(As, B_T)                                                    # :262-267
    T1 = xss.shape[0]
    flat = xss.reshape(T1, xss.shape[1], -1)
    sel = flat[torch.arange(T1, device=xss.device), Bs.long()]
    return sel.reshape((T1,) + tuple(xss.shape[2:])), Bs


def normalise(log_weights, log_space=False):
    """csmc.py:273-292."""
    return ops.normalise(log_weights, log_space=log_space)


def barker_move(key, ws):
    """csmc.py:295-297."""
    return ops.categorical(key, ws)
