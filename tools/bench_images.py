"""BASELINE configs 3-5 in shape (fbs_amd/image_configs.py): time per SMC step of gibbs_kernel over the image closures
with a randomly initialised UNet, split into network time (torch events around every network call) and sampler time
(everything else: resampling, fbsmi_em_concat / fbsmi_em_finish, normalisation, host loop).  One JSON line per
configuration.  Not the headline benchmark (bench.py), which carries the same legs with fewer steps."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd import image_configs, ops

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="c3,c4,c5")
ap.add_argument("--nsteps", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--rows", type=int, default=0, help="particles on this GPU (0: the configuration's per-GPU share)")
ap.add_argument("--find", type=int, default=0, help="1: torch.backends.cudnn.benchmark (MIOpen find mode)")
ap.add_argument("--chunk", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = bool(args.find)
for name in args.configs.split(","):
    c = image_configs.make(name, dev, dtype=args.dtype, nsteps=args.nsteps, chunk=args.chunk or None)
    n = args.rows or c.shard_rows
    image_configs.gibbs_sweep(c, ops.PRNGKey(3), n)
    torch.cuda.synchronize(); image_configs.network_ms(c)
    c.sb.profile = {}
    t0 = time.perf_counter(); image_configs.gibbs_sweep(c, ops.PRNGKey(4), n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    net_ms = image_configs.network_ms(c)
    pr = c.sb.profile
    ev = lambda a, b: sum(x.elapsed_time(y) for x, y in zip(pr[a], pr[b])) / max(1, len(pr[a]))
    T = args.nsteps
    print(json.dumps({"config": name, "workload": f"{c.cfg['task']} {c.cfg['image']}, UNet dim 64 (random init), {n} particles on this GPU "
                      f"(ensemble {c.cfg['nparticles']} over {c.cfg['ngpus']} GPU(s)), {T} of {image_configs.CONFIGS[name]['nsteps']} steps",
                      "dtype": args.dtype, "ms_per_step": dt / T * 1e3, "network_ms_per_step": net_ms / T,
                      "sampler_ms_per_step": (dt * 1e3 - net_ms) / T, "particle_steps_per_s": n * T / dt,
                      "concat_kernel_us": ev("concat0", "concat1") * 1e3, "finish_kernel_us": ev("finish0", "finish1") * 1e3,
                      "network_calls": c.timers["calls"], "find": args.find, "chunk": c.cfg["chunk"]}), flush=True)
