"""Bootstrap particle filter, backward smoother, particle-marginal MH (fbs/samplers/smc.py).

Closures are Python callables on GPU tensors as in the reference; resampling, gathers, logsumexp
and categorical draws are libfbsmi kernels.  Each function reproduces the reference's step order
(the three loops differ: see SURVEY.md section 3.3).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import ops
from ..score import bridge_of
from . import resampling as _resampling
from .common import MCMCState


def _fused_filter(transition_sampler, weight_closure, resampling, kwargs, nparticles):
    """(model, resampling name) when the fused analytic-model filter applies, else None."""
    model = getattr(transition_sampler, "_fbsmi_lg", None)
    if model is None or getattr(weight_closure, "_fbsmi_lg", None) is not model or kwargs:
        return None
    if getattr(weight_closure, "_role", "") != "likelihood_logpdf" or not model.fused_filter_supported(nparticles):
        return None
    name = "stratified" if resampling is _resampling.stratified else (
        "systematic" if resampling is _resampling.systematic else None)
    return (model, name) if name else None


def bootstrap_filter(transition_sampler, measurement_cond_pdf, vs, ts, init_sampler, key, nparticles, resampling,
                     log: bool = True, return_last: bool = True, **kwargs):
    """Bootstrap particle filter (smc.py:9-88) -> (samples, negative log-likelihood)."""
    nsteps = vs.shape[0] - 1
    fused = _fused_filter(transition_sampler, measurement_cond_pdf, resampling, kwargs, nparticles) if log else None
    if fused is not None and fused[0].T == nsteps:
        model, rname = fused
        key_init, _ = ops.split(key, 2)                                             # :77
        init_samples = init_sampler(key_init, vs[0], nparticles)                    # :78
        out = model.filter_handle(nparticles, "bootstrap", rname, store_path=not return_last).run(key, vs, init_samples)
        return (out[0].reshape(init_samples.shape), out[1]) if return_last else (
            out[2].reshape((nsteps + 1,) + tuple(init_samples.shape)), out[1])
    key_init, key_steps = ops.split(key, 2)                                         # :77
    us_prev = init_sampler(key_init, vs[0], nparticles)                             # :78
    keys = ops.split(key_steps, nsteps)                                             # :79
    log_nell = torch.zeros((), dtype=torch.float32, device=us_prev.device)
    logn = np.float32(math.log(nparticles))
    filtering = [us_prev]
    # closures of one ScoreBridge (log weights): one network evaluation per step, and with return_last the
    # resampling gather of step k is folded into the network-input kernel of step k + 1
    sb = bridge_of(transition_sampler, measurement_cond_pdf) if (log and set(kwargs) == {"mask_"}) else None
    pending = None                                                                  # ancestors not gathered yet
    for k in range(nsteps):                                                         # scan_body :58-74
        key_proposal, key_resampling = ops.split(keys[k], 2)
        v, v_prev, t_prev = vs[k + 1], vs[k], ts[k]
        if sb is not None:
            us, log_weights = sb.fused_step(us_prev, pending, v, v_prev, t_prev, key_proposal, kwargs["mask_"])
        else:
            us = transition_sampler(us_prev, v_prev, t_prev, key_proposal, **kwargs)    # :63
            log_weights = measurement_cond_pdf(v, us_prev, v_prev, t_prev, **kwargs)    # :65
        weights, c = ops.normalise(log_weights, log_space=False, return_lse=True)   # :66,68,69
        log_nell = log_nell - (c - logn)                                            # :67
        inds = resampling(weights, key_resampling)
        if sb is not None and return_last and k < nsteps - 1:
            us_prev, pending = us, inds
            continue
        us_prev, pending = ops.take_rows(us, inds), None                            # :72
        if not return_last:
            filtering.append(us_prev)
    if return_last:
        return us_prev, log_nell
    return torch.stack(filtering, 0), log_nell


def bootstrap_backward_smoother(key, filter_us, vs, ts, transition_logpdf, *args, **kwargs):
    """Backward particle smoother on bootstrap-filter output (smc.py:91-112)."""
    nsteps = filter_us.shape[0] - 1
    key_last, key_smoother = ops.split(key, 2)                                      # :108
    uT = ops.choice(key, filter_us[-1], axis=0)   # :109 -- the reference draws with the PARENT key
    keys = ops.split(key_smoother, nsteps)
    traj = [uT]
    u_kp1 = uT
    for s in range(nsteps):                       # filter_us[-2::-1], vs[-2::-1], ts[-2::-1]
        t = nsteps - 1 - s
        log_ws = transition_logpdf(u_kp1, filter_us[t], vs[t], ts[t], *args, **kwargs)  # :101
        w = ops.normalise(log_ws, log_space=False)                                  # :103
        i = ops.categorical(keys[s], w)                                             # :104
        u_kp1 = filter_us[t][i.long()]
        traj.append(u_kp1)
    return torch.stack(traj[::-1], 0)


def pmcmc_filter_step(key, vs_bridge, u0s, ts, transition_sampler, likelihood_logpdf, resampling, nparticles,
                      **kwargs):
    """Particle filter inside pMCMC (smc.py:115-158): weight -> resample old -> propagate."""
    nsteps = (ts.shape[0] if hasattr(ts, "shape") else len(ts)) - 1
    fused = _fused_filter(transition_sampler, likelihood_logpdf, resampling, kwargs, nparticles)
    if fused is not None and fused[0].T == nsteps:
        model, rname = fused
        uT, ell = model.filter_handle(nparticles, "pmcmc", rname).run(key, vs_bridge, u0s)
        return uT.reshape(u0s.shape), ell
    keys = ops.split(key, nsteps)                                                   # :154
    us = u0s
    log_ell = torch.zeros((), dtype=torch.float32, device=u0s.device)
    logn = np.float32(math.log(nparticles))
    sb = bridge_of(transition_sampler, likelihood_logpdf) if set(kwargs) == {"mask_"} else None
    for k in range(nsteps):                                                         # scan_body :138-152
        key_proposal, key_resampling = ops.split(keys[k], 2)
        v, v_prev, t_prev = vs_bridge[k + 1], vs_bridge[k], ts[k]
        if sb is not None:                                                          # :144-150, one network evaluation
            cell = {}

            def resample(lw):
                w, cell["c"] = ops.normalise(lw, log_space=False, return_lse=True)
                return resampling(w, key_resampling)

            us, _, _ = sb.fused_weight_then_propose(us, v, v_prev, t_prev, key_proposal, kwargs["mask_"], resample)
            log_ell = (log_ell - logn) + cell["c"]
            continue
        log_ws = likelihood_logpdf(v, us, v_prev, t_prev, **kwargs)                 # :144
        w, c = ops.normalise(log_ws, log_space=False, return_lse=True)              # :145,147,148
        log_ell = (log_ell - logn) + c                                              # :146
        inds = resampling(w, key_resampling)
        us_prev = ops.take_rows(us, inds)                                           # :149
        us = transition_sampler(us_prev, v_prev, t_prev, key_proposal, **kwargs)    # :150
    return us, log_ell


def pcn_proposal(key, delta: float, x, mean, sampler):
    """The pCN proposal (smc.py:161-168)."""
    beta = 2 / (2 + delta)
    key_rnds = ops.split(key, 2)
    rnds0, rnds1 = sampler(key_rnds[0]), sampler(key_rnds[1])
    p = x + math.sqrt(delta / 2) * (rnds0 - mean)
    return beta * p + (1 - beta) * mean + math.sqrt(1 - beta) * (rnds1 - mean)


def pmcmc_kernel(key, uT, log_ell, ys, y0, ts, fwd_ys_sampler, sde, ref_sampler, transition_sampler,
                 likelihood_logpdf, resampling, nparticles, delta: float = None, which_u: int = 0, **kwargs):
    """Particle-marginal MH kernel targeting p(uT | vT = y0) (smc.py:171-258).

    Returns (uT, log_ell, ys, MCMCState)."""
    key_prop, key_u0, key_filter, key_mh = ops.split(key, 4)                        # :231
    if delta is None:
        prop_ys = fwd_ys_sampler(key_prop, y0)                                      # :234
    else:
        ts_np = np.asarray(ts.detach().cpu() if isinstance(ts, torch.Tensor) else ts, np.float64).reshape(-1)
        y0_t = y0 if isinstance(y0, torch.Tensor) else torch.as_tensor(np.asarray(y0, np.float32), device=ys.device)
        coef = np.asarray(sde.mean(ts_np, ts_np[0], 1.0), np.float32)              # mean is linear in y0
        mean = torch.as_tensor(coef, device=ys.device).reshape((-1,) + (1,) * y0_t.dim()) * y0_t  # :236
        mean = mean.reshape(ys.shape)
        prop_ys = pcn_proposal(key_prop, delta, ys, mean, lambda key_: fwd_ys_sampler(key_, y0))  # :237
    vs = torch.flip(prop_ys, [0])                                                   # :239
    u0s = ref_sampler(key_u0, vs[0], nparticles)                                    # :241
    prop_uTs, prop_log_ell = pmcmc_filter_step(key_filter, vs, u0s, ts, transition_sampler, likelihood_logpdf,
                                               resampling, nparticles, **kwargs)    # :242
    prop_uT = prop_uTs[which_u]
    log_ell_t = log_ell if isinstance(log_ell, torch.Tensor) else torch.as_tensor(np.float32(log_ell),
                                                                                  device=prop_log_ell.device)
    log_acc_prob = torch.minimum(torch.zeros_like(prop_log_ell), prop_log_ell - log_ell_t)  # :246
    z = ops.uniform(key_mh, (), device=prop_log_ell.device)                         # :248
    acc_flag = ops.math_map("log", z.reshape(1)).reshape(()) < log_acc_prob        # :249
    state = MCMCState(acceptance_prob=torch.exp(log_acc_prob), is_accepted=acc_flag, prop_log_ell=prop_log_ell,
                      log_ell=log_ell_t)
    if bool(acc_flag.item()):                                                       # :255-258
        return prop_uT, prop_log_ell, prop_ys, state
    return uT, log_ell_t, ys, state


def twisted_smc(key, y, ts, init_sampler, transition_logpdf, twisting_logpdf, twisting_prop_sampler,
                twisting_prop_logpdf, resampling, nparticles, **kwargs):
    """Twisted SMC baseline (smc.py:261-309; Algorithm 1 of arXiv 2306.17775)."""
    nsteps = (ts.shape[0] if hasattr(ts, "shape") else len(ts)) - 1
    key_init, key_filter = ops.split(key, 2)
    keys = ops.split(key_filter, nsteps)
    xs = init_sampler(key_init, nparticles)
    log_ps = twisting_logpdf(y, xs, ts[0], **kwargs)
    log_ws = ops.normalise(log_ps, log_space=True)
    for k in range(nsteps):
        key_resampling, key_prop = ops.split(keys[k], 2)
        t_prev = ts[k + 1]  # the reference scans over ts[1:] (:306-307)
        inds = resampling(ops.math_map("exp", log_ws), key_resampling)
        xs_prev = ops.take_rows(xs, inds)
        log_ps_prev = log_ps[inds.long()]
        xs = twisting_prop_sampler(key_prop, xs_prev, t_prev, y, **kwargs)
        log_ps = twisting_logpdf(y, xs, t_prev, **kwargs)
        log_ws = (transition_logpdf(xs, xs_prev, t_prev) + log_ps
                  - twisting_prop_logpdf(xs, xs_prev, t_prev, y, **kwargs) - log_ps_prev)
        log_ws = ops.normalise(log_ws, log_space=True)
    return xs, log_ws
