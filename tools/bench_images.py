"""BASELINE config 3 in shape (MNIST 28x28 inpaint-15, UNet dim=64 pixel_shuffle, N particles): time
per SMC step of the closure tier with a randomly initialised score network, split into network time
and sampler time.  Not the headline benchmark (bench.py); prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd
from fbs_amd import ops
from fbs_amd.images import ImageRestore
from fbs_amd.score import ScoreBridge
from fbs_amd.sdes import StationaryLinLinearSDE
from fbs_amd.unet import UNet
from fbs_amd.samplers import gibbs_kernel

ap = argparse.ArgumentParser()
ap.add_argument("--nparticles", type=int, default=4096)
ap.add_argument("--nsteps", type=int, default=20)
ap.add_argument("--chunk", type=int, default=1024)
ap.add_argument("--dtype", default="fp32")
args = ap.parse_args()
dev = torch.device("cuda:0")
T, Tend = args.nsteps, 2.0
ts = np.linspace(0, Tend, T + 1)
sde = StationaryLinLinearSDE(0.02, 5.0, 0.0, Tend)
ds = ImageRestore("inpaint-15", (28, 28, 1), device=dev)
net = UNet(dt=Tend / 200, dim=64, in_channels=1, upsampling="pixel_shuffle").to(dev).eval()
tnet = {"s": 0.0, "n": 0}

def score_fn(x, t):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        if args.dtype == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x, t).float()
        else:
            out = net(x, t)
    e1.record()
    tnet.setdefault("ev", []).append((e0, e1))
    return out.reshape(x.shape)

sb = ScoreBridge(score_fn, ds, sde, ts, chunk=args.chunk)
mask = ds.gen_mask(ops.PRNGKey(1))
img = ops.uniform(ops.PRNGKey(2), (28, 28, 1), device=dev)
_, y0 = ds.unpack(img, mask)
x0 = torch.zeros(225, 1, device=dev)
bs = np.zeros(T + 1, np.int32)
N = args.nparticles
run = lambda k: gibbs_kernel(k, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, N, sb.transition_sampler,
                             sb.transition_logpdf, sb.likelihood_logpdf, mask_=mask)
run(ops.PRNGKey(3)); torch.cuda.synchronize(); tnet["ev"] = []
t0 = time.perf_counter(); run(ops.PRNGKey(4)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
net_s = sum(a.elapsed_time(b) for a, b in tnet["ev"]) / 1e3
print(json.dumps({"workload": f"MNIST inpaint-15, UNet dim=64 (random init), N={N}, T={T} (config 3 shape, fewer steps)",
                  "particle_steps_per_s": N * T / dt, "ms_per_step": dt / T * 1e3, "network_ms_per_step": net_s / T * 1e3,
                  "sampler_ms_per_step": (dt - net_s) / T * 1e3, "dtype": args.dtype, "unet_params": net.num_flat_params()}))
