"""BASELINE configs 3-5 in shape, on synthetic inputs: what experiments/imgs/inpainting.py,
supr.py and experiments/sb_imgs/supr.py set up before they call gibbs_kernel -- dataset geometry, SDE, time grid,
network, closures -- with a randomly initialised UNet (no checkpoints or datasets exist in this environment) and a
uniform[0, 1] test image.

  c3  MNIST 28x28 inpaint-15, UNet dim 64 pixel_shuffle, lin SDE beta in [0.02, 5], T = 2, 1000 steps, N = 4096,
      gibbs-eb-ef (experiments/imgs/inpainting.py:56-89, experiments/bashes/imgs_gibbs.sh:37)
  c4  MNIST Schrodinger-bridge supr-4 (sr_random = False), forward / backward drift UNets, T = 0.5, 50 steps,
      N = 8192 (2048 per GPU on 4), eb = ef = True (experiments/sb_imgs/supr.py:46-127,169-172)
  c5  CelebA-HQ 64x64x3 inpaint-32, same network family, T = 2, 1000 steps, N = 16 384 (2048 per GPU on 8)

`shard_rows` picks the per-GPU share of the ensemble (what one rank of fbs_amd.sharded owns).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .images import ImageRestore
from .score import ScoreBridge
from .sdes import StationaryLinLinearSDE
from .unet import UNet

CONFIGS = {
    "c3": dict(task="inpaint-15", image=(28, 28, 1), T=2.0, nsteps=1000, nparticles=4096, ngpus=1, mode="score",
               ef=True, chunk=4096),
    "c4": dict(task="supr-4", image=(28, 28, 1), T=0.5, nsteps=50, nparticles=8192, ngpus=4, mode="drift", ef=True,
               chunk=2048),
    "c5": dict(task="inpaint-32", image=(64, 64, 3), T=2.0, nsteps=1000, nparticles=16384, ngpus=8, mode="score",
               ef=True, chunk=2048),
}
# chunk = rows per network call: the whole shard in one power-of-two call plus explicit_final's one-row tail.  Power-of-two
# chunks keep MIOpen on kernels it ships (one call of 4097 / 2049 rows cost a fresh machine 50-120 s of kernel builds); with
# most of the network on this library's persistent-tile kernels bigger calls pay: config 3 43.7 / 39.9 / 38.3 ms per step at
# chunks of 1024 / 2048 / 4096, config 5's share 94.2 / 86.9 / 84.4 ms at 512 / 1024 / 2048 (tools/bench_images.py, one box).


def make(name: str, device, dtype: str = "bf16", nsteps: int | None = None, dim: int = 64, seed: int = 996,
         chunk: int | None = None, find: bool = False):
    """-> namespace(cfg, ds, sde, ts, net, sb, mask, y0, x0, shard_rows, closures, timers).
    find=True sets torch.backends.cudnn.benchmark (MIOpen's find mode): the convolutions of the fixed shapes of a sweep are
    tuned once per process -- minutes on first use, 13-15 % less network time per step afterwards (measured at configs 3, 5)."""
    if find:
        torch.backends.cudnn.benchmark = True
    cfg = dict(CONFIGS[name])
    if nsteps is not None:
        cfg["nsteps"] = int(nsteps)
    if chunk is not None:
        cfg["chunk"] = int(chunk)
    w, h, c = cfg["image"]
    ts = np.linspace(0.0, cfg["T"], cfg["nsteps"] + 1)
    sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=cfg["T"])       # inpainting.py:78
    ds = ImageRestore(cfg["task"], cfg["image"], sr_random=(cfg["mode"] != "drift"), device=device)
    torch.manual_seed(seed)
    mk = lambda: UNet(dt=cfg["T"] / 200, dim=dim, in_channels=c, upsampling="pixel_shuffle").to(device).eval()
    net = mk()
    net_fwd = mk() if cfg["mode"] == "drift" else None
    timers = {"events": [], "calls": 0}
    use_bf16 = dtype == "bf16"

    def run_net(module, x, t):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            if use_bf16:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = module(x, t)
            else:
                out = module(x, t)
        e1.record()
        timers["events"].append((e0, e1))
        timers["calls"] += 1
        return out

    score_fn = lambda x, t: run_net(net, x, t)
    fwd_drift = (lambda x, t: run_net(net_fwd, x, t).float().reshape(x.shape)) if net_fwd is not None else None
    sb = ScoreBridge(score_fn, ds, sde, ts, chunk=cfg["chunk"], mode=cfg["mode"],
                     net_input_dtype=torch.bfloat16 if use_bf16 else torch.float32, fwd_drift_fn=fwd_drift)
    key = ops.PRNGKey(seed)
    k_img, k_mask = ops.split(key, 2)
    img = ops.uniform(k_img, cfg["image"], device=device)
    mask = ds.gen_mask(k_mask)
    _, y0 = ds.unpack(img, mask)
    x0 = torch.zeros(ds.unobs_shape, dtype=torch.float32, device=device)
    return SimpleNamespace(name=name, cfg=cfg, ds=ds, sde=sde, ts=ts, net=net, net_fwd=net_fwd, sb=sb, mask=mask, y0=y0,
                           x0=x0, shard_rows=cfg["nparticles"] // cfg["ngpus"], timers=timers, dtype=dtype)


def network_ms(cfgobj) -> float:
    """Milliseconds the recorded network calls took on the device (call after torch.cuda.synchronize())."""
    tot = sum(a.elapsed_time(b) for a, b in cfgobj.timers["events"])
    cfgobj.timers["events"] = []
    return tot


def gibbs_sweep(cfgobj, key, nparticles: int, x0=None, bs_star=None):
    """One gibbs_kernel sweep (explicit_backward=True, explicit_final as the reference's script) of `nparticles`
    particles on this GPU.  -> (x0, us_star, bs_star, acc)."""
    from .samplers import gibbs_kernel
    c = cfgobj
    T = c.cfg["nsteps"]
    bs = np.zeros(T + 1, np.int32) if bs_star is None else bs_star
    sb = c.sb
    return gibbs_kernel(key, c.x0 if x0 is None else x0, c.y0, None, bs, c.ts, sb.fwd_sampler, c.sde, sb.unpack,
                        nparticles, sb.transition_sampler, sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False,
                        explicit_backward=True, explicit_final=c.cfg["ef"], mask_=c.mask)
