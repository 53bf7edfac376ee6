"""libfbsmi_dist's gather kernel (k_peer_gather: rows loaded through the window table, here the rank's own window) at the row sizes of
the sharded legs: algorithmic bytes (rows read + rows written) per second against the HBM peak.  One rank; over xGMI the remote
share of the rows is bounded by the links instead (7 x ~50 GB/s per direction and GPU)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd import ops  # noqa: E402
from fbs_amd.sharded import DistContext  # noqa: E402

dev = torch.device("cuda:0")
for R, shape in ((16385, (3072,)), (131072, (100,)), (1 << 22, (1,)), (1 << 22, (4,))):
    ctx = DistContext(R, device=dev)
    d = int(np.prod(shape))
    ctx.open_windows(d)
    rows = ops.normal(ops.PRNGKey(1), (R,) + shape, device=dev)
    A = torch.from_numpy(np.random.default_rng(0).integers(0, R, R).astype(np.int32)).to(dev)
    ctx.publish(rows)
    for _ in range(3):
        out = ctx.exchange(A, mode="peer", rowshape=shape)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = ctx.exchange(A, mode="peer", rowshape=shape)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    assert torch.equal(out, rows[A.long()])
    by = 2.0 * R * d * 4
    print(f"{R} rows x {d * 4} B (random ancestors): {dt * 1e6:8.1f} us per gather = {by / dt / 1e9:7.1f} GB/s algorithmic = {by / dt / 8e12:.3f} of the HBM peak")
    ctx.close()
