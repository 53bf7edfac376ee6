#!/usr/bin/env python3
"""Headline benchmark: particle-steps/sec of the forward-backward Gibbs sweep (BASELINE.json).

One "step" = one gibbs_kernel sweep (fbs/samplers/gibbs.py:68-168: explicit_backward=True,
explicit_final=False, marg_y=False, conditional killing resampling) of BASELINE config 2: the 2-D
joint Gaussian of tests/test_gibbs.py:24-39, N = 65 536 particles, T = 500 steps,
ts = linspace(0, 2, 501), analytic score.  A sweep is N*T particle-steps.  Inputs live on the
device; the whole sweep is a hipGraph replay (no host work inside the timed region beyond the
graph launches).

Multi-GPU (--gpus N under torch.distributed.run): one independent Gibbs chain per GPU -- the
reference's own parallel axis (nchains / --id replicas) -- no data-path collective, weak scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PARTICLES = 65536
T_STEPS = 500
T_END = 2.0
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNELS = ["norm", "cdf", "prop", "sumexp"]


def algorithmic_bytes_per_particle(du):
    """whole step 8*du + 24; Euler (prop) sub-sweep 8*du + 8."""
    return {"step": 8 * du + 24, "prop": 8 * du + 8, "norm": 8, "cdf": 8, "sumexp": 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nparticles", type=int, default=N_PARTICLES)
    ap.add_argument("--nsteps", type=int, default=T_STEPS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the sampler engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE

    N, T = args.nparticles, args.nsteps
    ts = np.linspace(0.0, T_END, T + 1)
    m0, cov0 = np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]])
    y0 = np.array([0.0], np.float32)
    br = fbs_amd.LinearGaussianBridge(m0, cov0, StationaryConstLinearSDE(a=-0.5, b=1.0), ts, du=1, device=dev)
    sweep = br.sweep_handle(N, True, False)

    key = fbs_amd.split(fbs_amd.PRNGKey(666), max(world, 2))[rank]  # one chain per rank
    x0 = np.zeros(1, np.float32)
    bs = np.zeros(T + 1, np.int32)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up (captures the graph on the first sweep)
    key, x0, bs, _ = sweep.chain(key, x0, y0, bs, max(args.warmup, 1), keep=False)
    sync()
    t0 = time.perf_counter()
    key, x0, bs, x0s = sweep.chain(key, x0, y0, bs, args.steps, keep=True)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    psteps = float(N) * T * args.steps * world
    value = psteps / dt
    ms_per_step = dt / args.steps * 1e3

    out = None
    if rank == 0:
        # ---- per-kernel durations with HIP events on the launch stream (untimed extra sweeps) ----
        sweep.profile(True)
        sweep.chain(key, x0, y0, bs, 2, keep=False, use_graph=False)
        torch.cuda.synchronize(dev)
        kern = {}
        for i, name in enumerate(KERNELS):
            us, n = sweep.kernel_us(i)
            kern[name] = {"avg_us": us, "launches": n}
        sweep.profile(False)
        bpp = algorithmic_bytes_per_particle(br.du)
        prop_bytes = bpp["prop"] * N
        prop_us = kern["prop"]["avg_us"]
        achieved = prop_bytes / (prop_us * 1e-6) / 1e9 if prop_us > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_lg_prop_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_lg_prop (gather + Euler-Maruyama + log-weight)",
                    "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                    "traffic": traffic, "bytes_per_launch": prop_bytes, "avg_launch_us": prop_us,
                    "timing": "hipEvent pairs around each launch on the launch stream (non-graph replay)",
                    "whole_sweep_GBps": bpp["step"] * float(N) * T / (ms_per_step * 1e-3) / 1e9,
                    "kernels_us": {k: v["avg_us"] for k, v in kern.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle as O
            h = br.host
            om = O.LGModel(br.du, br.dv, br.dt, h["G"], h["g"], h["sd"], h["lognorm"], h["F"], h["sqQ"])
            c0 = time.perf_counter()
            O.bench_gibbs_lg(om, 666, np.zeros(1, np.float32), y0, N, args.cpu_sweeps)
            cdt = time.perf_counter() - c0
            cpu = {"value": float(N) * T * args.cpu_sweeps / cdt, "unit": "particle-steps/s", "cores": 1,
                   "kind": "port", "sample": f"{args.cpu_sweeps} sweeps of the same workload (N={N}, T={T}) on the "
                   f"single-threaded C oracle, {cdt:.1f} s",
                   "note": "CPU restatement of the reference algorithm (not JAX: JAX is not installable here)"}
        out = {"metric": "particle-steps/sec (N x T per Gibbs sweep)", "value": value, "unit": "particle-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"2-D Gaussian bridge toy (BASELINE config 2): N={N} particles, T={T} steps, "
                          "ts=linspace(0,2), analytic score, gibbs_kernel eb=True ef=False marg_y=False, "
                          "conditional killing resampling; one step = one Gibbs sweep",
                          "nparticles": N, "nsteps": T, "du": br.du, "dv": br.dv,
                          "parallelism": f"{world} independent chain(s), one per GPU"},
               "roofline": roofline, "cpu_baseline": cpu,
               "x0_mean_of_timed_sweeps": float(x0s.float().mean().item())}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
