"""CPU, world_size 2, gloo: the multi-process plumbing bench.py uses for N > 1 -- independent key
chains per rank and the max-over-ranks timing reduction (no data-path collective: chains are
independent units, weak scaling)."""
import os
import socket
import sys

import numpy as np
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    key = bench.rank_key(world, rank)
    t = bench.max_over_ranks(0.1 * (rank + 1), dist, torch.device("cpu"))
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(key.astype(np.int64)))
    dist.barrier()
    if rank == 0:
        q.put((t, [g.tolist() for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, keys = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(t - 0.2) < 1e-9                      # MAX over ranks
    assert keys[0] != keys[1]                       # independent chains
    sys.path.insert(0, ROOT)
    import oracle as O
    want = O.split(O.PRNGKey(666), 2)
    assert keys == [[int(a), int(b)] for a, b in want]
