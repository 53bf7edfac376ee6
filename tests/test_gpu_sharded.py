"""GPU, 2 and 3 ranks on ONE card (gloo rendezvous, tensors staged through the host -- the production backend is
nccl = RCCL, which a one-GPU box cannot run with more than one rank): the PRODUCT's sharded path end to end --
libfbsmi kernels for every local operation, the fused score-network step on each rank's rows with its slice of the
noise, both ancestor exchanges, explicit_final's ragged N + 1 rows -- must return on every rank exactly what the
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
            for exchange in ("all_gather", "all_to_all"):
                sh = sharded.ParticleShards(n + (1 if ef else 0), dist=dist, exchange=exchange)
                got = sharded.gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n, sb.transition_sampler,
                                           sb.transition_logpdf, sb.likelihood_logpdf, sh, explicit_final=ef, mask_=mask)
                same = all(torch.equal(a, b) for a, b in zip(got, want))
                ok = ok and same and sh.bytes_moved > 0
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
