#!/usr/bin/env python3
"""Headline benchmark: particle-steps/sec of the forward-backward Gibbs sweep (BASELINE.json).

One "step" = one gibbs_kernel sweep (fbs/samplers/gibbs.py:68-168: explicit_backward=True,
explicit_final=False, marg_y=False, conditional killing resampling) of BASELINE config 2: the 2-D
joint Gaussian of tests/test_gibbs.py:24-39, N = 65 536 particles, T = 500 steps,
ts = linspace(0, 2, 501), analytic score.  A sweep is N*T particle-steps.  Inputs live on the
device; the whole sweep is a hipGraph replay (no host work inside the timed region beyond the
graph launches).

The batch of one step is `--nchains` independent chains (default 4 = the reference driver's default,
experiments/toy/gp_gibbs.py:25, which vmaps gibbs_kernel over chains :172-173).  A batch of four or more chains is
driven as two groups of half the chains, each group's launches on its own stream (fbs_amd.LGSweep): the step kernels are
latency-bound, the groups' launches interleave (+6 % at 4 chains, +18-27 % at 16-32; FBSMI_CHAIN_GROUPS=1 keeps one
group) -- same results bit for bit.  The single-chain rate is measured too and reported in `single_chain`.

Multi-GPU (--gpus N under torch.distributed.run): every GPU runs its own batch of chains -- the
reference's own parallel axes (nchains, --id replicas) -- no data-path collective, weak scaling: that is `value`.
The SAME run also times one particle ensemble SHARDED over the N ranks (`sharded_c5`: BASELINE config 5's
16 384 particles, fbs_amd/sharded.py: RCCL all_gather of the log-weights + ancestor rows by all_gather or by
all_to_all), strong scaling, so a 1 -> 8 GPU series carries both axes.

Extra objects of the line (all measured in this process unless they say otherwise): `single_chain`, `batch_scan`,
`spill` (a working set beyond the Infinity Cache), `gp100` (the reference's d = 100 toy), `c3` / `c4_shard` /
`c5_shard` (the image configurations per SMC step, network and sampler time apart), `em_finish` (roofline of the
fused score-network step kernel at config 5's per-GPU shape), `sharded_c5`.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PARTICLES = 65536
T_STEPS = 500
T_END = 2.0
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNELS = ["norm", "cdf", "prop"]


def algorithmic_bytes_per_particle(du):
    """whole step 8*du + 24; Euler (prop) sub-sweep 8*du + 8."""
    return {"step": 8 * du + 24, "prop": 8 * du + 8, "norm": 8, "cdf": 8}


def rank_key(world, rank):
    """An independent threefry key chain per rank: split(PRNGKey(666), max(world, 2))[rank]."""
    import fbs_amd
    return fbs_amd.split(fbs_amd.PRNGKey(666), max(world, 2))[rank]


def max_over_ranks(dt, dist, dev):
    """The timed region's duration is the slowest rank's."""
    if dist is None:
        return dt
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def gp100_leg(dev):
    """The reference's own toy experiment (experiments/bashes/toy_gibbs.sh + tabulators/tabulate_toy.py:
    gp_gibbs.py --d=100 --nparticles=100 --explicit_backward, 4 chains, T = 200): joint dimension D = 200, so
    the affine drift is a (slots x 200) x (200 x 200) product per step and runs on the f32 matrix cores
    (k_lgw_gemm).  Outside the timed region of the headline metric; reported beside it."""
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    d, T, C = 100, 200, 4
    zs = np.linspace(0., 5., d)
    cov = np.exp(-np.abs(zs[None, :] - zs[:, None]))                               # gp_gibbs.py:39-40
    joint = np.block([[cov, cov], [cov, cov + np.eye(d)]])                          # :55-57
    ts = np.linspace(0., 1., T + 1)
    br = fbs_amd.LinearGaussianBridge(np.zeros(2 * d), joint, StationaryConstLinearSDE(a=-0.5, b=1.), ts, d, device=dev)
    y0 = np.random.default_rng(0).normal(size=d).astype(np.float32)
    out = []
    for N, nrep in ((100, 20), (10000, 5), (100000, 3)):
        sw = br.sweep_handle(N, True, False, nchains=C)
        k, x, b, _ = sw.chain(fbs_amd.PRNGKey(1), np.zeros((C, d), np.float32), y0, np.zeros((C, T + 1), np.int32), 2,
                              keep=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sw.chain(k, x, y0, b, nrep, keep=False)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / nrep
        tf = 2.0 * N * (2 * d) ** 2 * T * C / dt / 1e12
        out.append({"nparticles": N, "d": d, "nsteps": T, "nchains": C, "ms_per_sweep": dt * 1e3,
                    "value": float(N) * T * C / dt, "unit": "particle-steps/s",
                    "drift_TFLOPs": tf, "frac_of_f32_mfma_peak": tf / 157.3})
        del sw
        torch.cuda.empty_cache()
    return {"workload": "gp_gibbs.py --d=100 --explicit_backward (toy_gibbs.sh), 4 chains; nparticles=100 is the paper's "
                        "table (tabulate_toy.py:16), 10000 and 100000 show the drift kernel loaded", "runs": out,
            "note": "f32 MFMA (exact fmaf chain) dense peak 157.3 TFLOP/s; at nparticles=100 a step is one launch "
                    "and latency-bound; from 10000 particles up a step is five launches around k_lgw_gemm_fat"}


def image_legs(dev, nsteps, dtypes=("f32", "bf16")):
    """Configs 3-5 in shape (fbs_amd/image_configs.py: synthetic image, random-init UNet dim 64): one gibbs_kernel sweep of
    `nsteps` of the configuration's steps on this GPU's share of the particles, network time (torch events around every
    network call) and sampler time (everything else) apart.  Two arithmetic types: **float32** is the configuration's number --
    the reference's UNet computes in float32 (fbs/nn/unet.py:85-86, x64 off) -- and bf16 autocast (network input written,
    output read in bfloat16 by the step kernels) is this build's fast path, reported beside it under `<label>` / `<label>_f32`.
    `ms_per_step` = the timed sweep divided by its steps (it carries the sweep's fixed parts: with 6 steps, explicit_final's
    initial weights alone are a seventh network evaluation); `ms_per_step_marginal` = what one more step costs (bf16 legs)."""
    from fbs_amd import image_configs, ops
    out = {}
    for dtype in dtypes:
        ns = nsteps if dtype == "bf16" else max(1, min(nsteps, 3))    # the float32 sweeps are ~4x as long: fewer steps
        for name, label in (("c3", "c3"), ("c4", "c4_shard"), ("c5", "c5_shard")):
            c = image_configs.make(name, dev, dtype=dtype, nsteps=ns)
            n = c.shard_rows
            image_configs.gibbs_sweep(c, ops.PRNGKey(3), n)
            torch.cuda.synchronize(dev)
            image_configs.network_ms(c)
            c.sb.profile = {}
            t0 = time.perf_counter()
            image_configs.gibbs_sweep(c, ops.PRNGKey(4), n)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            net_ms = image_configs.network_ms(c)
            pr = c.sb.profile
            ev = lambda a, b: sum(x.elapsed_time(y) for x, y in zip(pr[a], pr[b])) / max(1, len(pr[a])) * 1e3
            full = image_configs.CONFIGS[name]
            # a sweep has fixed parts (explicit_final's initial weights are one more network evaluation, the forward path, the
            # final move): the MARGINAL step is the difference between two sweeps of different length
            marginal = None
            if ns >= 4 and dtype == "bf16":
                ns2 = max(1, ns // 3)
                c2 = image_configs.make(name, dev, dtype=dtype, nsteps=ns2)
                image_configs.gibbs_sweep(c2, ops.PRNGKey(3), n)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                image_configs.gibbs_sweep(c2, ops.PRNGKey(4), n)
                torch.cuda.synchronize(dev)
                marginal = (dt - (time.perf_counter() - t1)) / (ns - ns2) * 1e3
                del c2
            out[label + ("_f32" if dtype == "f32" else "")] = {
                "workload": f"{full['task']} on {full['image']}, UNet dim 64 random init, {n} particles on this GPU "
                            f"(+1: explicit_final; ensemble {full['nparticles']} over {full['ngpus']} GPU(s)), "
                            f"{ns} of the configuration's {full['nsteps']} steps timed",
                "dtype": "f32 (the reference's precision: this is the configuration's figure)" if dtype == "f32"
                         else "bf16 autocast (fast path; narrower than the reference's float32 network)",
                "ms_per_step": dt / ns * 1e3, "network_ms_per_step": net_ms / ns,
                "sampler_ms_per_step": (dt * 1e3 - net_ms) / ns, "particle_steps_per_s": n * ns / dt,
                "ms_per_step_marginal": marginal,
                "full_sweep_s_extrapolated": (dt + (marginal if marginal else dt / ns * 1e3) * 1e-3 * (full["nsteps"] - ns)),
                "concat_kernel_us_with_event_overhead": ev("concat0", "concat1"),
                "finish_kernel_us_with_event_overhead": ev("finish0", "finish1")}
            del c
            torch.cuda.empty_cache()
    return out


def em_finish_roofline(dev):
    """The fused score-network step kernel (fbsmi_em_finish: ancestor gather + unpack + Euler-Maruyama with in-kernel
    normal + pin + row-summed log-density) at config 5's per-GPU shape (2048 rows of du = 3072, dv = 9216), float32
    network output: back-to-back launches between one hipEvent pair on the launch stream, cycling through buffer sets
    larger than the Infinity Cache so that every launch streams from HBM.  Algorithmic bytes per particle: 8 du + 8
    (Euler sub-sweep) + 4 (du + dv) (the network output it consumes)."""
    from fbs_amd import _lib, ops
    from fbs_amd.images import ImageRestore
    from fbs_amd.score import EMMask
    n, shape = 2048, (64, 64, 3)
    ds = ImageRestore("inpaint-32", shape, device=dev)
    em = EMMask(ds.gen_mask(ops.PRNGKey(1)), 3, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    nsets = 4
    sets = [dict(us=torch.randn((n, em.du), device=dev, generator=g), net=torch.randn((n, em.D), device=dev, generator=g),
                 A=torch.randint(0, n, (n,), device=dev, generator=g, dtype=torch.int32),
                 us_new=torch.empty((n, em.du), device=dev), lw=torch.empty(n, device=dev),
                 img=torch.empty((n, em.D), device=dev)) for _ in range(nsets)]
    for b in sets:   # the production dtype of configs 3-5: the autocast network reads and writes bfloat16
        b["net16"] = b["net"].to(torch.bfloat16)
        b["img16"] = torch.empty((n, em.D), device=dev, dtype=torch.bfloat16)
    v, vp = torch.randn(em.dv, device=dev, generator=g), torch.randn(em.dv, device=dev, generator=g)
    pin = torch.randn(em.du, device=dev, generator=g)
    st = torch.cuda.current_stream().cuda_stream

    def finish(b, dt16=False):
        _lib.call("fbsmi_em_finish", em.ref, b["us"].data_ptr(), b["A"].data_ptr(), b["net16" if dt16 else "net"].data_ptr(), None,
                  1 if dt16 else 0, 0, 1.3, 2.6, 0.002, 0.0721, v.data_ptr(), vp.data_ptr(), 1, 2, n, 0, n, 5, pin.data_ptr(),
                  b["us_new"].data_ptr(), b["lw"].data_ptr(), st)

    def concat(b, dt16=False):
        _lib.call("fbsmi_em_concat", em.ref, b["us"].data_ptr(), b["A"].data_ptr(), vp.data_ptr(), n, 1 if dt16 else 0,
                  b["img16" if dt16 else "img"].data_ptr(), st)

    def timed(fn, iters=10):
        for b in sets:
            fn(b)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            for b in sets:
                fn(b)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / (iters * nsets) * 1e3

    fus, cus = timed(finish), timed(concat)
    fus16, cus16 = timed(lambda b: finish(b, True)), timed(lambda b: concat(b, True))
    fbytes, cbytes = n * (8 * em.du + 8 + 4 * em.D), n * (4 * em.du + 4 * em.D)
    fbytes16, cbytes16 = n * (8 * em.du + 8 + 2 * em.D), n * (4 * em.du + 2 * em.D)
    roof = lambda by, us: {"avg_launch_us": us, "bytes_per_launch": by, "achieved": by / us / 1e3, "frac": by / us / 1e3 / PEAK_HBM_GBS}
    return {"bound": "hbm", "kernel": "k_em_finish (fbs_amd/csrc/fbsmi_em.hip)", "shape": "config 5 per-GPU share: 2048 rows, "
            "du=3072, dv=9216, float32 network output, cold (4 buffer sets, 600 MB)", "avg_launch_us": fus,
            "bytes_per_launch": fbytes, "achieved": fbytes / fus / 1e3, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": fbytes / fus / 1e3 / PEAK_HBM_GBS, "traffic": None,
            "concat_kernel": roof(cbytes, cus),
            "bf16_network": {"note": "the same launches with the bfloat16 network output / input the autocast networks of configs "
                                     "3-5 actually produce and take (2 instead of 4 bytes per network value)",
                             "finish_kernel": roof(fbytes16, fus16), "concat_kernel": roof(cbytes16, cus16)},
            "limiter": "vector-instruction issue: rocprofv3 (profiles/r02_em_pmc.json) counts 22.4 M wave-instructions per launch -- "
                       "Threefry-2x32/20 + erf_inv for 6.3 M normals, the row log-densities with a correctly rounded division each -- "
                       "i.e. 21.8 k per SIMD x ~3 clocks = 27-28 us at 2.4 GHz before any memory wait, and 45 % of the wave-cycles "
                       "are issue stalls (SQ_WAIT_INST_ANY): 0.55 of the HBM roof (23 us for the bfloat16 shape) lies below that floor",
            "note": "the same box's torch copy_ of 100 MB cold buffers moves 5.1 TB/s = 0.64 of the 8 TB/s peak; the finish kernel's "
                    "proposal and log-density roles queue on the same memory pipeline and their times add (with the normals "
                    "pre-drawn it is no faster): see DESIGN.md section 5.0"}


def note(rank, msg):
    """Progress on stderr (rank 0): a long multi-leg run must not look hung to whoever is watching it."""
    if rank == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def sharded_leg(dev, dist, world, rank, nsteps):
    """BASELINE config 5 as ONE ensemble of 16 384 particles sharded over the ranks (fbs_amd/sharded.py): per SMC step an
    all_gather of the log-weights, the ancestor rows by all_gather or all_to_all over RCCL or loaded from their owners' windows
    (exchange="peer": libfbsmi_dist, hipIpc over xGMI), the network and the fused step kernels on the local rows.  Strong scaling: the ensemble is fixed, the per-rank share is 16384 / world."""
    from fbs_amd import image_configs, ops, sharded
    c = image_configs.make("c5", dev, dtype="bf16", nsteps=nsteps)
    N = c.cfg["nparticles"]
    sb = c.sb
    bs = np.zeros(nsteps + 1, np.int32)
    res = {}
    # one rank: nothing is exchanged, the three forms are the same run
    for exchange in (("all_gather", "all_to_all", "peer") if world > 1 else ("all_gather",)):
        sh = sharded.ParticleShards(N + 1, dist=dist, exchange=exchange)
        run = lambda key: sharded.gibbs_kernel(key, c.x0, c.y0, None, bs, c.ts, sb.fwd_sampler, c.sde, sb.unpack, N,
                                               sb.transition_sampler, sb.transition_logpdf, sb.likelihood_logpdf, sh,
                                               explicit_final=True, mask_=c.mask)
        run(ops.PRNGKey(5))
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        image_configs.network_ms(c)
        sh.bytes_moved = 0
        t0 = time.perf_counter()
        out = run(ops.PRNGKey(6))
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        dt = max_over_ranks(time.perf_counter() - t0, dist, dev)
        net_ms = image_configs.network_ms(c)
        sh.close()
        note(rank, f"sharded_c5 {exchange}: {dt / nsteps * 1e3:.1f} ms per step")
        res[exchange] = {"value": float(N) * nsteps / dt, "unit": "particle-steps/s", "ms_per_step": dt / nsteps * 1e3,
                         "network_ms_per_step_rank0": net_ms / nsteps, "rows_per_rank": sh.n,
                         "ancestor_exchange_bytes_received_per_step_rank0": sh.bytes_moved / nsteps,
                         "logweight_all_gather_bytes_per_step": 4 * (sh.world - 1) * sh.n,
                         "x0_checksum": float(out[0].double().sum().item())}
    res["workload"] = (f"CelebA-64 inpaint-32 (config 5), ONE ensemble of {N} (+1) particles over {world} rank(s), UNet dim 64 "
                       f"random init bf16, gibbs_kernel eb=ef=True, {nsteps} of 1000 steps timed; strong scaling")
    res["scaling"] = "strong"
    res["n_gpus"] = world
    return res


def sharded_lg_leg(dev, dist, world, rank, nsteps):
    """north_star's split on the Gaussian-bridge workload: ONE ensemble sharded over the ranks (fbs_amd/sharded.py: contiguous
    slot ranges per rank, the log-weights all_gathered, the ancestor rows by all_gather -- rows of 4 / 400 bytes --, the
    propagation and weighting on the local rows through the fbsmi_lg_*_rows kernels, the noise a row slice of the global
    draw).  Two ensembles: BASELINE config 2's model at N = 2^22 (du = 1) and the reference's d = 100 toy at N = 131 072.
    Strong scaling; `nsteps` SMC steps per timed sweep.  Also returns the per-step ESS / log-normaliser diagnostics."""
    import fbs_amd
    from fbs_amd import ops, sharded
    from fbs_amd.sdes import StationaryConstLinearSDE
    out = {}
    d = 100
    zs = np.linspace(0., 5., d)
    cov = np.exp(-np.abs(zs[None, :] - zs[:, None]))
    cases = {"toy2d_N4194304": (np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]), 1, 1 << 22, T_END),
             "gp100_N131072": (np.zeros(2 * d), np.block([[cov, cov], [cov, cov + np.eye(d)]]), d, 131072, 1.0)}
    for name, (m0, cov0, du, N, tend) in cases.items():
        ts = np.linspace(0.0, tend, nsteps + 1)
        br = fbs_amd.LinearGaussianBridge(m0, cov0, StationaryConstLinearSDE(a=-0.5, b=1.0), ts, du=du, device=dev)
        y0 = torch.zeros(br.dv, device=dev)
        x0 = torch.zeros(du, device=dev)
        bs = np.zeros(nsteps + 1, np.int32)
        def timed(exchange):
            sh = sharded.ParticleShards(N, dist=dist, exchange=exchange)
            run = lambda key: sharded.gibbs_kernel(key, x0, y0, None, bs, ts, br.fwd_sampler, br.sde, br.unpack, N,
                                                   br.transition_sampler, br.transition_logpdf, br.likelihood_logpdf, sh)
            run(ops.PRNGKey(7))
            torch.cuda.synchronize(dev)
            if dist is not None:
                dist.barrier()
            sh.bytes_moved = 0
            t0 = time.perf_counter()
            res = run(ops.PRNGKey(8))
            torch.cuda.synchronize(dev)
            if dist is not None:
                dist.barrier()
            dt = max_over_ranks(time.perf_counter() - t0, dist, dev)
            diag = sh.diagnostics.cpu().numpy()
            moved = sh.bytes_moved
            sh.close()
            note(rank, f"sharded_lg {name} {exchange}: {dt / nsteps * 1e3:.2f} ms per step")
            return {"value": float(N) * nsteps / dt, "unit": "particle-steps/s", "ms_per_step": dt / nsteps * 1e3,
                    "nparticles": N, "du": du, "rows_per_rank": sh.n, "exchange": sh.exchange_for(torch.empty(1, du)),
                    "ancestor_exchange_bytes_received_per_step_rank0": moved / nsteps,
                    "logweight_all_gather_bytes_per_step": 4 * (sh.world - 1) * sh.n,
                    "ess_last_step": float(diag[-1, 1]), "log_normaliser_sum": float(diag[:, 0].sum()),
                    "x0_checksum": float(res[0].double().sum().item())}

        out[name] = timed("auto")
        if world > 1:                                  # the same ensemble with the rows loaded from their owners' windows
            p = timed("peer")
            out[name]["peer"] = {k: p[k] for k in ("value", "ms_per_step", "ancestor_exchange_bytes_received_per_step_rank0",
                                                   "x0_checksum")}
        del br
        torch.cuda.empty_cache()
    out["workload"] = (f"one linear-Gaussian ensemble over {world} rank(s), closure tier + fbs_amd/sharded.py, gibbs_kernel eb=True "
                       f"ef=False, {nsteps} SMC steps timed; strong scaling (host loop: ~15 launches and two collectives per step)")
    out["scaling"] = "strong"
    out["n_gpus"] = world
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nparticles", type=int, default=N_PARTICLES)
    ap.add_argument("--nsteps", type=int, default=T_STEPS)
    ap.add_argument("--nchains", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=2)
    ap.add_argument("--no-single-chain", action="store_true")
    ap.add_argument("--no-gp100", action="store_true", help="skip the d = 100 Gaussian-process toy leg (`gp100`)")
    ap.add_argument("--batch-scan", type=str, default="16,32",
                    help="extra chain-batch sizes timed (untimed region) and reported in `batch_scan`; '' to skip")
    ap.add_argument("--image-steps", type=int, default=6, help="SMC steps timed per image configuration (0: skip the legs)")
    ap.add_argument("--image-dtype", choices=["both", "f32", "bf16"], default="both",
                    help="arithmetic of the image legs' network: f32 = the reference's (the configurations' figures), bf16 = "
                         "autocast fast path; both by default")
    ap.add_argument("--sharded-steps", type=int, default=10, help="SMC steps of the sharded config-5 ensemble (0: skip)")
    ap.add_argument("--sharded-timeout", type=float, default=900.0,
                    help="seconds the sharded legs may take on more than one rank before the line is printed without them")
    ap.add_argument("--sharded-lg-steps", type=int, default=20,
                    help="SMC steps of the sharded linear-Gaussian ensembles (`sharded_lg`; 0: skip)")
    ap.add_argument("--no-spill", action="store_true", help="skip the beyond-the-Infinity-Cache leg (`spill`)")
    args = ap.parse_args()

    # native pieces are built before anything touches the GPU (hipcc / gcc children must not inherit a profiler's preload)
    from fbs_amd import _lib as _fbsmi_lib
    _fbsmi_lib.build_dist()   # libfbsmi, then the exchange library that links it
    if not args.no_cpu_baseline:
        import oracle as _oracle_build
        _oracle_build.build()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the sampler engine has no CPU path")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL; FBSMI_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on one GPU
        backend = os.environ.get("FBSMI_BENCH_BACKEND", "nccl")
        import datetime
        tmo = datetime.timedelta(seconds=300)   # a rank that dies inside a collective must not hold the others for the default 10 min
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE

    N, T = args.nparticles, args.nsteps
    ts = np.linspace(0.0, T_END, T + 1)
    m0, cov0 = np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]])
    y0 = np.array([0.0], np.float32)
    br = fbs_amd.LinearGaussianBridge(m0, cov0, StationaryConstLinearSDE(a=-0.5, b=1.0), ts, du=1, device=dev)
    C = args.nchains
    sweep = br.sweep_handle(N, True, False, nchains=C)

    key = rank_key(world, rank)  # an independent key chain per rank
    x0 = np.zeros((C, 1), np.float32)
    bs = np.zeros((C, T + 1), np.int32)

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
    dt = max_over_ranks(dt, dist, dev)
    psteps = float(N) * T * C * args.steps * world
    value = psteps / dt
    ms_per_step = dt / args.steps * 1e3

    out = None
    if rank == 0:
        # ---- per-kernel durations with HIP events on the launch stream (untimed extra sweeps) ----
        # A batch of >= 4 chains runs as two groups of half the chains on two streams (fbs_amd.LGSweep): the launches that
        # exist are one group's, so one group is profiled, alone, and bytes per launch count ITS chains.
        ph = sweep.children[0] if sweep.children else sweep
        Cp = ph.C
        kp, xp, bp_ = fbs_amd.PRNGKey(11), np.zeros((Cp, 1), np.float32), np.zeros((Cp, T + 1), np.int32)
        kp, xp, bp_, _ = ph.chain(kp, xp, y0, bp_, 2, keep=False)
        torch.cuda.synchronize(dev)
        tp0 = time.perf_counter()
        nrep_p = max(4, args.steps // 2)
        kp, xp, bp_, _ = ph.chain(kp, xp, y0, bp_, nrep_p, keep=False)
        torch.cuda.synchronize(dev)
        group_step_us = (time.perf_counter() - tp0) / nrep_p * 1e6 / T     # graph-timed step of one group running alone
        ph.profile(True)
        ph.chain(kp, xp, y0, bp_, 2, keep=False, use_graph=False)
        torch.cuda.synchronize(dev)
        kern = {}
        for i, name in enumerate(KERNELS):
            us, n = ph.kernel_us(i)
            if n:
                kern[name] = {"avg_us": us, "launches": n}
        ph.profile(False)
        bpp = algorithmic_bytes_per_particle(br.du)
        prop_bytes = bpp["prop"] * N * Cp
        # A hipEvent pair brackets each launch, so every per-kernel figure carries the same additive
        # event overhead c.  The step kernels tile a step of the graph-timed region, hence
        # c = (sum of their event figures - graph-timed step) / their number; durations below are net of c.
        raw = {k: v["avg_us"] for k, v in kern.items()}
        step_us = group_step_us
        c_ev = max(0.0, (sum(raw.values()) - step_us) / float(len(raw)))
        net = {k: max(v - c_ev, 1e-3) for k, v in raw.items()}
        prop_us = net["prop"]
        achieved = prop_bytes / (prop_us * 1e-6) / 1e9
        # PMC figures cannot be collected inside this process: they come from a rocprofv3 --pmc run of this same command
        # whose summary is committed under profiles/ (newest round first), and are reported ONLY under `from_profile_file`,
        # and only when that file describes the kernel and workload this run launched.
        prof = None
        kname = "k_lg_prop1t" if "cdf" not in kern else "k_lg_prop"
        for rr in range(9, 0, -1):
            tpath = os.path.join(ROOT, "profiles", f"r{rr:02d}_pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                except Exception:
                    break
                wl = tj.get("workload", {})
                if tj.get("kernel", "").startswith(kname) and wl == {"nparticles": N, "nsteps": T, "nchains": C} and \
                        tj.get("chains_per_launch", C) == Cp:
                    insts = tj.get("valu_insts_per_launch")
                    prof = {"file": os.path.relpath(tpath, ROOT), "kernel": tj.get("kernel"), "workload": wl,
                            "hbm_bytes_per_launch": tj.get("bytes_per_launch"),
                            "valu_insts_per_launch": insts,
                            "valu_issue_frac": (insts * float(tj.get("cycles_per_valu_inst", 2.7)) / 4.0 /
                                                (256.0 * 2.4e3 * prop_us)) if insts else None,
                            "note": "measured by rocprofv3 in a separate run of this command, not in this process; "
                                    "valu_issue_frac = instructions x measured issue cycles / (1024 SIMDs x 2.4 GHz x launch time)"}
                break
        roofline = {"bound": "hbm", "kernel": "k_lg_prop1t (tree-walking searches + resample + gather + Euler-Maruyama + log-weight)"
                    if "cdf" not in kern else "k_lg_prop (resample + gather + Euler-Maruyama + log-weight)",
                    "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                    "traffic": None, "from_profile_file": prof,
                    # the roof that does bind once a few chains share a CU, as its own fraction (rocprofv3 instruction counts of a
                    # separate run of this command x measured issue cycles, against this run's launch time)
                    "valu_roof": ({"bound": "vector-instruction issue", "unit": "wave-instructions/s",
                                   "achieved": prof["valu_insts_per_launch"] / (prop_us * 1e-6),
                                   "peak": 1024 * 2.4e9 / 2.7, "frac": prof["valu_issue_frac"],
                                   "note": "1024 SIMDs x 2.4 GHz / 2.7 measured issue cycles per instruction of this mix; one "
                                           "launch running alone (two are in flight during the timed sweeps)"}
                                  if prof and prof.get("valu_issue_frac") is not None else None),
                    "limiter": "vector-instruction issue and dependent round trips (the per-step working set is cache resident): "
                               "the HBM fraction is reported because SURVEY 8(d) assigns this path the HBM roof, not because "
                               "the kernel is near it",
                    "bytes_per_launch": prop_bytes, "avg_launch_us": prop_us, "chains_per_launch": Cp,
                    "chain_groups": len(sweep.children) if sweep.children else 1,
                    "timing": "hipEvent pairs around each launch on the launch stream of ONE chain group running alone, net of "
                              "the event overhead calibrated against that group's graph-timed step (see bench.py)",
                    "event_overhead_us": c_ev, "raw_event_us": raw, "kernels_us": net,
                    "whole_sweep_GBps": bpp["step"] * float(N) * T * C / (ms_per_step * 1e-3) / 1e9,
                    "whole_step_frac": bpp["step"] * float(N) * T * C / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "launches_per_step": len(kern),
                    "note": "working set per step is a few MB (cache-resident) and the kernel is latency- then "
                            "VALU-bound (in-kernel Threefry + erf_inv), not HBM-bound: see DESIGN.md.  With chain_groups = 2 the "
                            "timed sweeps keep TWO such launches in flight (one per group, each carrying chains_per_launch chains): "
                            "kernel durations then add up to more than the wall clock, and achieved / frac describe one launch "
                            "running alone"}
        single = None
        if C != 1 and not args.no_single_chain:
            sw1 = br.sweep_handle(N, True, False, nchains=1)
            k1, x1, b1, _ = sw1.chain(key, np.zeros(1, np.float32), y0, np.zeros(T + 1, np.int32), 2, keep=False)
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            nrep = max(3, min(args.steps, 10))
            sw1.chain(k1, x1, y0, b1, nrep, keep=False)
            torch.cuda.synchronize(dev)
            sdt = (time.perf_counter() - s0) / nrep
            single = {"value": float(N) * T / sdt, "unit": "particle-steps/s", "ms_per_sweep": sdt * 1e3,
                      "note": "same workload with nchains=1 on this GPU (latency-bound: one chain cannot fill the chip)"}
        scan = None
        if world == 1 and args.batch_scan and not args.no_single_chain:
            scan = []
            for cb in [int(x) for x in args.batch_scan.split(",") if x]:
                swb = br.sweep_handle(N, True, False, nchains=cb)
                kb, xb, bb, _ = swb.chain(key, np.zeros((cb, 1), np.float32), y0, np.zeros((cb, T + 1), np.int32), 1,
                                          keep=False)
                torch.cuda.synchronize(dev)
                b0 = time.perf_counter()
                swb.chain(kb, xb, y0, bb, 3, keep=False)
                torch.cuda.synchronize(dev)
                bdt = (time.perf_counter() - b0) / 3
                scan.append({"nchains": cb, "value": float(N) * T * cb / bdt, "ms_per_sweep": bdt * 1e3,
                             "whole_sweep_GBps": bpp["step"] * float(N) * T * cb / bdt / 1e9})
                del swb
        spill = None
        if world == 1 and not args.no_spill and not args.no_single_chain:
            # a working set beyond the 256 MiB Infinity Cache: 4 chains x 2^22 particles, ~100 MB of per-step arrays each
            Ns, Ts, Cs = 1 << 22, 20, 4
            brs = fbs_amd.LinearGaussianBridge(m0, cov0, StationaryConstLinearSDE(a=-0.5, b=1.0),
                                               np.linspace(0.0, T_END, Ts + 1), du=1, device=dev)
            sws = brs.sweep_handle(Ns, True, False, nchains=Cs)
            ks, xs_, bs_, _ = sws.chain(key, np.zeros((Cs, 1), np.float32), y0, np.zeros((Cs, Ts + 1), np.int32), 1, keep=False)
            torch.cuda.synchronize(dev)
            q0 = time.perf_counter()
            sws.chain(ks, xs_, y0, bs_, 3, keep=False)
            torch.cuda.synchronize(dev)
            qdt = (time.perf_counter() - q0) / 3
            spill = {"nparticles": Ns, "nsteps": Ts, "nchains": Cs, "value": float(Ns) * Ts * Cs / qdt,
                     "unit": "particle-steps/s", "ms_per_sweep": qdt * 1e3,
                     "whole_sweep_GBps": bpp["step"] * float(Ns) * Ts * Cs / qdt / 1e9,
                     "whole_step_frac_of_hbm_peak": bpp["step"] * float(Ns) * Ts * Cs / qdt / 1e9 / PEAK_HBM_GBS,
                     "note": "same model and kernels as the headline, per-step arrays ~400 MB: beyond the Infinity Cache"}
            del sws, brs
        gp100 = None
        if world == 1 and not args.no_gp100 and not args.no_single_chain:
            gp100 = gp100_leg(dev)
        images, emroof = None, None
        if world == 1 and args.image_steps > 0 and not args.no_single_chain:
            emroof = em_finish_roofline(dev)
            images = image_legs(dev, args.image_steps, ("f32", "bf16") if args.image_dtype == "both" else (args.image_dtype,))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle as O
            h = br.host
            om = O.LGModel(br.du, br.dv, br.dt, h["G"], h["g"], h["sd"], h["lognorm"], h["F"], h["sqQ"])
            nthreads = max(1, min(os.cpu_count() or 1, 16))
            # bounded sample: one probing sweep sizes the timed run to about 15 s of CPU work
            p0 = time.perf_counter()
            O.bench_gibbs_lg(om, 665, np.zeros(1, np.float32), y0, N, 1, threads=nthreads)
            probe = time.perf_counter() - p0
            sweeps = max(args.cpu_sweeps, min(64, int(15.0 / max(probe, 1e-3))))
            c0 = time.perf_counter()
            _, used = O.bench_gibbs_lg(om, 666, np.zeros(1, np.float32), y0, N, sweeps, threads=nthreads)
            cdt = time.perf_counter() - c0
            cpu = {"value": float(N) * T * sweeps / cdt, "unit": "particle-steps/s", "cores": used,
                   "kind": "port", "sample": f"{sweeps} single-chain sweeps of the same workload (N={N}, T={T}) on the "
                   f"C oracle with OpenMP over independent particle loops, {used} host thread(s), {cdt:.1f} s",
                   "note": "CPU restatement of the reference algorithm (not JAX: JAX is not installable here)"}
        out = {"metric": "particle-steps/sec (N x T per Gibbs sweep)", "value": value, "unit": "particle-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"2-D Gaussian bridge toy (BASELINE config 2): N={N} particles, T={T} steps, "
                          "ts=linspace(0,2), analytic score, gibbs_kernel eb=True ef=False marg_y=False, "
                          f"conditional killing resampling; one step = one Gibbs sweep of a batch of {C} chain(s) "
                          "(reference driver default nchains=4, vmapped)",
                          "nparticles": N, "nsteps": T, "nchains": C, "du": br.du, "dv": br.dv,
                          "parallelism": f"{world} GPU(s) x {C} independent chain(s) each, no collective"},
               "roofline": roofline, "cpu_baseline": cpu,
               "value_single_chain": single["value"] if single else None, "single_chain": single, "batch_scan": scan,
               "spill": spill, "gp100": gp100, "em_finish": emroof,
               "x0_mean_of_timed_sweeps": float(x0s.float().mean().item())}
        if images:
            out.update(images)
    # every rank takes part in the sharded ensemble (collectives); rank 0 reports.  The headline above is already measured:
    # should an exchange hang on some node, the line is still printed (the sharded legs marked as timed out) and the job ends.
    def give_up():
        if rank == 0:
            out.setdefault("sharded_c5", {"error": f"no result within {args.sharded_timeout} s"})
            out.setdefault("sharded_lg", {"error": f"no result within {args.sharded_timeout} s"})
            print(json.dumps(out), flush=True)
        os._exit(0)

    note(rank, "headline and single-GPU legs done; sharded legs next")
    watchdog = threading.Timer(args.sharded_timeout, give_up)
    watchdog.daemon = True
    if world > 1:
        watchdog.start()
    shard = None
    if args.sharded_steps > 0 and not args.no_single_chain:
        del sweep
        torch.cuda.empty_cache()
        try:
            shard = sharded_leg(dev, dist, world, rank, args.sharded_steps)
        except Exception as e:  # every rank runs the same deterministic code: a failure is reported, the headline stays
            shard = {"error": f"{type(e).__name__}: {e}"}
            print(f"[bench rank {rank}] sharded_c5 failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    shard_lg = None
    if args.sharded_lg_steps > 0 and not args.no_single_chain:
        try:
            shard_lg = sharded_lg_leg(dev, dist, world, rank, args.sharded_lg_steps)
        except Exception as e:
            shard_lg = {"error": f"{type(e).__name__}: {e}"}
            print(f"[bench rank {rank}] sharded_lg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    if rank == 0:
        out["sharded_c5"] = shard
        out["sharded_lg"] = shard_lg
    if dist is not None:
        try:   # (a rank that failed inside a sharded leg may have left its peers in a collective: the line is printed regardless)
            dist.barrier()
        except Exception as e:
            print(f"[bench rank {rank}] final barrier: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
