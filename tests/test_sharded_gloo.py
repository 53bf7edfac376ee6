"""CPU, gloo, world_size 2 and 4: the sharded-ensemble path (fbs_amd/sharded.py).  The exchange
logic (all_gather of log-weights, locally derived all_to_all plan for ancestor rows, broadcast of
the selected particle) is the product's; the LOCAL array operations are injected from the CPU
oracle (the product's own local backend is GPU-only).  The sharded Gibbs sweep must return, on every
rank, exactly what the unsharded oracle sweep returns."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_backend(O):
    from types import SimpleNamespace
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def set_row(dst, row, val):
        out = dst.clone()
        out[row] = val
        return out

    def force_move(key, w, k):
        i, a = O.force_move(key, w.numpy(), k)
        return torch.tensor(i), torch.tensor(a)

    return SimpleNamespace(split=O.split, normalise=lambda x, log_space=False: t(O.normalise(x.numpy(), log_space)),
                           exp=lambda x: t(O.exp(x.numpy())), take_rows=lambda src, idx: src[idx.long()],
                           set_row=set_row,
                           cond_resampling=lambda key, w, i, j, c: t(O.cond_killing(key, w.numpy(), i, j, c)),
                           force_move=force_move,
                           randint=lambda key, shape, lo, hi, device=None: t(O.randint(key, shape, lo, hi)))


def _worker(rank, world, port, N, T, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from fbs_amd import sharded
    from helpers import toy_4d
    toy = toy_4d()
    du = toy["du"]
    ts = np.linspace(0, 1, T + 1)
    m = O.make_lg(toy["m0"], toy["cov0"], O.sde_const(-0.5, 1.0), ts, du)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sh = sharded.ParticleShards(N, dist=dist)

    # exchange plan on a random ancestor array: every rank must end up with us_full[A[my slots]]
    rng = np.random.default_rng(123)
    A = t(rng.integers(0, N, N).astype(np.int32))
    us_full = t(rng.normal(size=(N, 3)).astype(np.float32))
    got = sh.gather_ancestors(us_full[sh.offset:sh.offset + sh.n].clone(), A, lambda s, i: s[i.long()])
    ok_plan = torch.equal(got, us_full[A.long()][sh.offset:sh.offset + sh.n])
    ok_gather = torch.equal(sh.all_gather_vec(us_full[sh.offset:sh.offset + sh.n].clone()), us_full)

    def transition_sampler(us_prev, v_prev, t_prev, key, row_slice=None):
        off, cnt, tot = row_slice
        k = int(np.argmin(np.abs(ts[:-1] - float(t_prev))))
        full = np.zeros((tot, du), np.float32)
        full[off:off + cnt] = us_prev.numpy()
        return t(O.lg_transition_sampler(m, k, full, v_prev.numpy(), key)[off:off + cnt])

    def likelihood_logpdf(v, us_prev, v_prev, t_prev):
        k = int(np.argmin(np.abs(ts[:-1] - float(t_prev))))
        return t(O.lg_likelihood_logpdf(m, k, v.numpy(), us_prev.numpy(), v_prev.numpy()))

    fwd_sampler = lambda key, x0, y0: t(O.lg_fwd_sampler(m, key, np.concatenate([x0.numpy(), y0.numpy()])))
    unpack = lambda xy: (xy[..., :du], xy[..., du:])
    x0 = rng.normal(size=du).astype(np.float32)
    bs = rng.integers(0, N, T + 1).astype(np.int32)
    key = O.PRNGKey(2024)
    out = sharded.gibbs_kernel(key, t(x0), t(toy["y0"]), None, bs, ts, fwd_sampler, None, unpack, N,
                               transition_sampler, None, likelihood_logpdf, sh, _oracle_backend(O))
    want = O.gibbs_kernel_lg(m, key, x0, toy["y0"], bs, N, True, False)
    same = all(np.array_equal(np.asarray(a.numpy()).view(np.uint8) if a.dtype != torch.bool else a.numpy(),
                              np.ascontiguousarray(b).view(np.uint8) if b.dtype != np.bool_ else b)
               for a, b in zip(out, want))
    dist.barrier()
    q.put((rank, bool(ok_plan), bool(ok_gather), bool(same)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_gibbs_equals_unsharded(world):
    N, T = 64, 12
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_plan, ok_gather, same in res:
        assert ok_plan, f"rank {rank}: ancestor exchange wrong"
        assert ok_gather, f"rank {rank}: all_gather wrong"
        assert same, f"rank {rank}: sharded sweep differs from the unsharded oracle sweep"


def test_shards_reject_uneven_split():
    sys.path.insert(0, ROOT)
    from fbs_amd import sharded
    sh = sharded.ParticleShards(10)
    assert (sh.world, sh.rank, sh.n, sh.offset) == (1, 0, 10, 0) and sh.owns(9) and not sh.owns(10)

    class FakeDist:
        def is_initialized(self): return True
        def get_world_size(self, g): return 3
        def get_rank(self, g): return 1
    with pytest.raises(ValueError):
        sharded.ParticleShards(10, dist=FakeDist())
