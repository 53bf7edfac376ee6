"""GPU, 2 and 3 ranks on ONE card (gloo rendezvous, tensors staged through the host -- the production backend is
nccl = RCCL, which a one-GPU box cannot run with more than one rank): the PRODUCT's sharded path end to end --
libfbsmi kernels for every local operation, the fused score-network step on each rank's rows with its slice of the
noise, the three ancestor exchanges (the collective two, and libfbsmi_dist's peer windows: every rank's rows in a window the
other processes map through hipIpc, rows loaded straight from their owner), explicit_final's ragged N + 1 rows -- must return on every rank exactly what the
unsharded gibbs_kernel returns on one GPU.  The network is a deterministic elementwise stand-in, so the comparison is
bit for bit (a real UNet's kernels may depend on the batch composition)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_net(x, t):
    y = x.float() * 0.75
    return y + (0.1 * float(t))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fbs_amd import ops, sharded
        from fbs_amd.images import ImageRestore
        from fbs_amd.samplers import gibbs_kernel
        from fbs_amd.score import ScoreBridge
        from fbs_amd.sdes import StationaryLinLinearSDE
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        T, n, Tend = 7, 50, 2.0
        ts = np.linspace(0, Tend, T + 1)
        sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=Tend)
        ds = ImageRestore("inpaint-8", (16, 16, 3), device=dev)
        sb = ScoreBridge(_toy_net, ds, sde, ts, chunk=16)
        mask = ds.gen_mask(ops.PRNGKey(12))
        img = ops.uniform(ops.PRNGKey(13), (16, 16, 3), device=dev)
        _, y0 = ds.unpack(img, mask)
        x0 = torch.zeros(ds.unobs_shape, device=dev)
        bs = np.random.default_rng(3).integers(0, n, T + 1).astype(np.int32)
        key = ops.PRNGKey(14)
        ok = True
        for ef in (False, True):
            want = gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n, sb.transition_sampler,
                                sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False, explicit_backward=True,
                                explicit_final=ef, mask_=mask)
            for exchange in ("all_gather", "all_to_all", "peer"):
                sh = sharded.ParticleShards(n + (1 if ef else 0), dist=dist, exchange=exchange)
                got = sharded.gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n, sb.transition_sampler,
                                           sb.transition_logpdf, sb.likelihood_logpdf, sh, explicit_final=ef, mask_=mask)
                same = all(torch.equal(a, b) for a, b in zip(got, want))
                ok = ok and same and sh.bytes_moved > 0
                sh.close()                                           # (collective for "peer": nobody reads a freed window)
        # ---- the linear-Gaussian closures (fbsmi_lg_transition_sampler_rows: the noise a row slice of the global draw) on a
        # sharded ensemble against the fused single-GPU sweep engine, for a narrow model and for a wide (d = 24) one
        import fbs_amd
        from fbs_amd.sdes import StationaryConstLinearSDE
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import toy_4d, toy_gp
        for toy, N, T2 in ((toy_4d(), 300, 9), (toy_gp(24), 130, 5)):
            ts2 = np.linspace(0, 1.0, T2 + 1)
            br = fbs_amd.LinearGaussianBridge(toy["m0"], toy["cov0"], StationaryConstLinearSDE(a=-0.5, b=1.0), ts2, toy["du"], device=dev)
            rng = np.random.default_rng(N)
            x0l = torch.from_numpy(rng.normal(size=br.du).astype(np.float32)).to(dev)
            y0l = torch.from_numpy(np.asarray(toy["y0"], np.float32)).to(dev)
            bsl = rng.integers(0, N, T2 + 1).astype(np.int32)
            keyl = ops.PRNGKey(77)
            want = br.gibbs_kernel(keyl, x0l, y0l, bsl, N, True, False)
            for exchange in ("all_gather", "all_to_all", "auto", "peer"):
                sh = sharded.ParticleShards(N, dist=dist, exchange=exchange)
                got = sharded.gibbs_kernel(keyl, x0l, y0l, None, bsl, ts2, br.fwd_sampler, br.sde, br.unpack, N,
                                           br.transition_sampler, br.transition_logpdf, br.likelihood_logpdf, sh)
                same = all(torch.equal(a.reshape(b.shape), b) for a, b in zip(got, want))
                dg = sh.diagnostics                                  # (T + 1, 2): log-normaliser increment, ESS
                okd = dg is not None and tuple(dg.shape) == (T2 + 1, 2) and bool(torch.isfinite(dg).all()) and \
                    bool((dg[:, 1] >= 1.0 - 1e-4).all()) and bool((dg[:, 1] <= N + 1e-2).all())
                ok = ok and same and okd
                sh.close()
        dist.barrier()
        q.put((rank, bool(ok), ""))
    except Exception as e:  # report instead of hanging the other ranks' collectives forever
        q.put((rank, False, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_product_sharded_sweep_equals_unsharded(world, dev):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, f"rank {rank}: sharded sweep differs from the unsharded one {err}"
