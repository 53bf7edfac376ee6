"""2 ranks on ONE card (gloo), config-5 image geometry, a deterministic elementwise stand-in network: the sharded product path
(both exchanges) against the unsharded gibbs_kernel, bit for bit, at a size where a step moves hundreds of MB.
python tools/sharded_bigcheck.py [rows] [steps]"""
import os, socket, sys
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def toy(x, t):
    return x.float() * 0.75 + 0.1 * float(t)

def worker(rank, world, port, n, T, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fbs_amd import ops, sharded
        from fbs_amd.images import ImageRestore
        from fbs_amd.samplers import gibbs_kernel
        from fbs_amd.score import ScoreBridge
        from fbs_amd.sdes import StationaryLinLinearSDE
        dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
        ts = np.linspace(0, 2.0, T + 1)
        sde = StationaryLinLinearSDE(beta_min=0.02, beta_max=5.0, t0=0.0, T=2.0)
        ds = ImageRestore("inpaint-32", (64, 64, 3), device=dev)
        sb = ScoreBridge(toy, ds, sde, ts, chunk=512)
        mask = ds.gen_mask(ops.PRNGKey(12))
        _, y0 = ds.unpack(ops.uniform(ops.PRNGKey(13), (64, 64, 3), device=dev), mask)
        x0 = torch.zeros(ds.unobs_shape, device=dev)
        bs = np.zeros(T + 1, np.int32)
        key = ops.PRNGKey(14)
        want = gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n, sb.transition_sampler,
                            sb.transition_logpdf, sb.likelihood_logpdf, marg_y=False, explicit_backward=True,
                            explicit_final=True, mask_=mask)
        msg = []
        for exchange in ("all_gather", "all_to_all"):
            for rep in range(2):
                sh = sharded.ParticleShards(n + 1, dist=dist, exchange=exchange)
                got = sharded.gibbs_kernel(key, x0, y0, None, bs, ts, sb.fwd_sampler, sde, sb.unpack, n, sb.transition_sampler,
                                           sb.transition_logpdf, sb.likelihood_logpdf, sh, explicit_final=True, mask_=mask)
                msg.append((exchange, rep, all(torch.equal(a, b) for a, b in zip(got, want))))
        dist.barrier()
        q.put((rank, msg))
    except Exception as e:
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, n, T, q)) for r in range(2)]
    [p.start() for p in ps]
    for _ in range(2): print(q.get(timeout=900))
    [p.join(60) for p in ps]
