"""CPU, gloo, world_size 2, 3 and 4: the sharded-ensemble path (fbs_amd/sharded.py).  The exchange logic
(all_gather of log-weights; ancestor rows by all_gather or by the device-derived all_to_all plan; ragged last
shard; noise drawn as row slices of the global draw; broadcast of the selected particle) is the product's; the
LOCAL array operations are injected from the CPU oracle (the product's own local backend is GPU-only).  The
sharded sweep must return, on every rank, exactly what the unsharded oracle sweep returns -- for the
linear-Gaussian closures (closure by closure) and for the image closures (the fused step around one network
evaluation, explicit_final's N + 1 rows)."""
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

    def normal(key, shape, device=None, rows=None):
        z = O.normal(key, shape)
        return t(z if rows is None else z[rows[0]:rows[0] + rows[1]])

    return SimpleNamespace(split=O.split, normalise=lambda x, log_space=False: t(O.normalise(x.numpy(), log_space)),
                           exp=lambda x: t(O.exp(x.numpy())), take_rows=lambda src, idx: src[idx.long()],
                           set_row=set_row,
                           cond_resampling=lambda key, w, i, j, c: t(O.cond_killing(key, w.numpy(), i, j, c)),
                           force_move=force_move, normal=normal,
                           randint=lambda key, shape, lo, hi, device=None: t(O.randint(key, shape, lo, hi)))


def _image_case(O, sharded, sh_of, world):
    """The image closures on a sharded ensemble of N + 1 rows (explicit_final): the fused step of one rank =
    oracle/em.py on its rows with its slice of the noise; must equal the unsharded oracle forward pass."""
    from oracle import em, images
    shape = (12, 12, 2)
    _, unobs, obs = images.gen_inpaint_mask(O.PRNGKey(2), shape, 5, 5)
    u_off, v_off, role = em.element_tables(unobs, obs, 2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    rng = np.random.default_rng(4)
    T, n = 5, 4 * world + 1                                   # one row more than the ranks divide
    du, dv = u_off.size, v_off.size
    us_star = rng.normal(size=(T + 1, du)).astype(np.float32)
    vs = rng.normal(size=(T + 1, dv)).astype(np.float32)
    bs = rng.integers(0, n, T + 1).astype(np.int32)
    us0 = rng.normal(size=(n, du)).astype(np.float32)
    lw0 = rng.normal(size=n).astype(np.float32)
    ts = np.linspace(0, 2, T + 1)
    coef = lambda tp: (0, 0.3 + 0.1 * float(tp), 1.1, 0.2)
    netf = lambda img, tp: ((img * np.float32(0.5)).astype(np.float32) + np.float32(0.01 * float(tp))).astype(np.float32)
    key = O.PRNGKey(31)
    wantA, wantlw, wantus = em.forward_pass(key, us_star, bs, vs, us0, lw0, ts, coef, netf, 0.4, u_off, v_off, role)
    ok = True
    for exchange in ("all_gather", "all_to_all"):
        sh = sh_of(n, exchange)
        be = _oracle_backend(O)

        def fused_step(closures, us_src, A_local, v, v_prev, t_prev, key_, pin, row_slice, kwargs):
            off, cnt, tot = row_slice
            A = None if A_local is None else A_local.numpy()
            img = em.concat(us_src.numpy(), A, v_prev.numpy(), role)
            mode, cx, cs, sd = coef(t_prev)
            pr, pv = (-1, None) if pin is None else (pin[0], pin[1].numpy())
            un, lw = em.finish(us_src.numpy(), A, netf(img, t_prev), mode, np.float32(cx), np.float32(cs), np.float32(0.4),
                               np.float32(sd), v.numpy(), v_prev.numpy(), key_, tot, off, pr, pv, u_off, v_off)
            return t(un), t(lw)

        be.fused_step = fused_step
        init_sampler = lambda k_, n_, row_slice=None: t(us0[row_slice[0]:row_slice[0] + row_slice[1]])
        init_ll = lambda v0, u0s, v1, row_slice=None: t(lw0[row_slice[0]:row_slice[0] + row_slice[1]])
        lw, us = sharded.forward_pass(key, t(us_star), bs, t(vs), ts, init_sampler, init_ll, None, None, n, sh, be)
        ok = ok and np.array_equal(lw.numpy().view(np.uint32), wantlw.view(np.uint32))
        ok = ok and np.array_equal(us.numpy().view(np.uint32), wantus[sh.offset:sh.offset + sh.count].view(np.uint32))
        ok = ok and (world == 1 or sh.bytes_moved > 0)
    return bool(ok)


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
    sh = sharded.ParticleShards(N, dist=dist, exchange="all_to_all")

    # exchange plan on random ancestor arrays, also with a ragged last shard: every rank must end up with us_full[A[my slots]]
    rng = np.random.default_rng(123)
    ok_plan = ok_gather = True
    for R in (N, N + 1, N - 3):
        shr = sharded.ParticleShards(R, dist=dist, exchange="all_to_all")
        A = t(rng.integers(0, R, R).astype(np.int32))
        us_full = t(rng.normal(size=(R, 3)).astype(np.float32))
        mine = us_full[shr.offset:shr.offset + shr.count].clone()
        got = shr.gather_ancestors(mine, A, lambda s, i: s[i.long()])
        ok_plan = ok_plan and torch.equal(got, us_full[A.long()][shr.offset:shr.offset + shr.count])
        ok_gather = ok_gather and torch.equal(shr.all_gather_rows(mine), us_full)
    # a collapsed ensemble: every ancestor is one row of rank 0, so every other rank sends NOTHING (zero-row send buffers); and
    # its mirror image, everything from the last rank
    for a_all in (1, N - 1):
        A = t(np.full(N, a_all, np.int32))
        us_full = t(rng.normal(size=(N, 2, 3)).astype(np.float32))
        mine = us_full[sh.offset:sh.offset + sh.count].clone()
        got = sh.gather_ancestors(mine, A, lambda s, i: s[i.long()])
        ok_plan = ok_plan and torch.equal(got, us_full[A.long()][sh.offset:sh.offset + sh.count])

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
    same = True
    want = O.gibbs_kernel_lg(m, key, x0, toy["y0"], bs, N, True, False)
    for exchange in ("all_to_all", "all_gather"):
        she = sharded.ParticleShards(N, dist=dist, exchange=exchange)
        out = sharded.gibbs_kernel(key, t(x0), t(toy["y0"]), None, bs, ts, fwd_sampler, None, unpack, N,
                                   transition_sampler, None, likelihood_logpdf, she, _oracle_backend(O))
        same = same and all(np.array_equal(np.asarray(a.numpy()).view(np.uint8) if a.dtype != torch.bool else a.numpy(),
                                           np.ascontiguousarray(b).view(np.uint8) if b.dtype != np.bool_ else b)
                            for a, b in zip(out, want))
    same_img = _image_case(O, sharded, lambda R, ex: sharded.ParticleShards(R, dist=dist, exchange=ex), world)
    dist.barrier()
    q.put((rank, bool(ok_plan), bool(ok_gather), bool(same and same_img)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
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


def test_shards_ragged_split_and_argument_checks():
    sys.path.insert(0, ROOT)
    from fbs_amd import sharded
    sh = sharded.ParticleShards(10)
    assert (sh.world, sh.rank, sh.n, sh.offset, sh.count) == (1, 0, 10, 0, 10) and sh.owns(9) and not sh.owns(10)

    class FakeDist:
        def __init__(self, world, rank): self.w, self.r = world, rank
        def is_initialized(self): return True
        def get_world_size(self, g): return self.w
        def get_rank(self, g): return self.r
        def get_backend(self, g=None): return "gloo"
    # 10 rows over 3 ranks: shards of 4, 4 and 2 rows
    got = [(s.offset, s.count) for s in (sharded.ParticleShards(10, dist=FakeDist(3, r)) for r in range(3))]
    assert got == [(0, 4), (4, 4), (8, 2)]
    with pytest.raises(ValueError):
        sharded.ParticleShards(2, dist=FakeDist(3, 1))
    with pytest.raises(ValueError):
        sharded.ParticleShards(10, exchange="ring")
    # 9 rows over 8 ranks would leave the last three ranks without rows (shards of 2): refused, not crashed on later
    with pytest.raises(ValueError, match="would own no row"):
        sharded.ParticleShards(9, dist=FakeDist(8, 7))
    # "auto": by row size and world size
    import torch
    sh3 = sharded.ParticleShards(30, dist=FakeDist(3, 0), exchange="auto")
    assert sh3.exchange_for(torch.zeros(10, 2048)) == "all_to_all" and sh3.exchange_for(torch.zeros(10, 4)) == "all_gather"
    assert sharded.ParticleShards(30, dist=FakeDist(2, 0), exchange="auto").exchange_for(torch.zeros(15, 2048)) == "all_gather"
