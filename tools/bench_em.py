"""Times fbsmi_em_concat / fbsmi_em_finish (fbs_amd/csrc/fbsmi_em.hip) at the per-step shapes of BASELINE
configs 3-5, with hipEvents on the launch stream, and prints one JSON line per shape with the algorithmic
bytes per launch and the fraction of the 8 TB/s HBM peak.  Algorithmic bytes of the finish kernel per particle
(Euler sub-sweep 8 du + 8, plus the network output it consumes): 8 du + 8 + s (du + dv),
s = 4 (float32 network output) or 2 (bfloat16).  Not the headline benchmark (bench.py)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd import _lib, ops  # noqa: E402
from fbs_amd.images import ImageRestore  # noqa: E402
from fbs_amd.score import EMMask  # noqa: E402

SHAPES = {
    "c3": ("inpaint-15", (28, 28, 1), 4096, 4096),
    "c3_ef": ("inpaint-15", (28, 28, 1), 4097, 4097),
    "c4_shard": ("supr-4", (28, 28, 1), 2048, 8192),
    "c5_shard": ("inpaint-32", (64, 64, 3), 2048, 16384),
    "c5_whole": ("inpaint-32", (64, 64, 3), 16384, 16384),
}


ONLY = set()


def time_kernel(fn, nsets, iters):
    """Average duration in microseconds of fn(set) over iters * nsets back-to-back launches between ONE hipEvent pair
    (host launch latency overlaps the previous kernel), cycling through `nsets` disjoint buffer sets so that a launch
    never finds its operands in the 256 MiB Infinity Cache."""
    for s in range(nsets):
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for s in range(nsets):
            fn(s)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * nsets) * 1e3


def run(name, net_dtype, sliced, iters, nsets):
    task, shape, n, n_total = SHAPES[name]
    dev = torch.device("cuda:0")
    ds = ImageRestore(task, shape, sr_random=False, device=dev)
    mask = ds.gen_mask(ops.PRNGKey(1))
    em = EMMask(mask, shape[2], dev)
    g = torch.Generator(device=dev).manual_seed(0)
    ndt = 0 if net_dtype == torch.float32 else 1
    sets = []
    for _ in range(nsets):
        sets.append(dict(us=torch.randn((n, em.du), device=dev, generator=g),
                         net=torch.randn((n, em.D), device=dev, generator=g).to(net_dtype),
                         A=torch.randint(0, n, (n,), device=dev, generator=g, dtype=torch.int32),
                         us_new=torch.empty((n, em.du), device=dev), lw=torch.empty(n, device=dev),
                         img=torch.empty((n, em.D), device=dev, dtype=net_dtype)))
    v, vp = torch.randn(em.dv, device=dev, generator=g), torch.randn(em.dv, device=dev, generator=g)
    pin = torch.randn(em.du, device=dev, generator=g)
    row0, ntot = ((n_total // n // 2) * n, n_total) if sliced else (0, n)
    st = torch.cuda.current_stream().cuda_stream

    def finish(i):
        b = sets[i]
        _lib.call("fbsmi_em_finish", em.ref, b["us"].data_ptr(), b["A"].data_ptr(), b["net"].data_ptr(), None, ndt, 0, 1.3,
                  2.6, 0.002, 0.0721, v.data_ptr(), vp.data_ptr(), 1, 2, ntot, row0, n, 5, pin.data_ptr(),
                  b["us_new"].data_ptr(), b["lw"].data_ptr(), st)

    def concat(i):
        b = sets[i]
        _lib.call("fbsmi_em_concat", em.ref, b["us"].data_ptr(), b["A"].data_ptr(), vp.data_ptr(), n, ndt,
                  b["img"].data_ptr(), st)

    def prop_only(i):
        b = sets[i]
        _lib.call("fbsmi_em_finish", em.ref, b["us"].data_ptr(), b["A"].data_ptr(), b["net"].data_ptr(), None, ndt, 0, 1.3,
                  2.6, 0.002, 0.0721, None, None, 1, 2, ntot, row0, n, 5, pin.data_ptr(), b["us_new"].data_ptr(), None, st)

    def lw_only(i):
        b = sets[i]
        _lib.call("fbsmi_em_finish", em.ref, None, None, b["net"].data_ptr(), None, ndt, 0, 1.3, 2.6, 0.002, 0.0721,
                  v.data_ptr(), vp.data_ptr(), 1, 2, ntot, row0, n, -1, None, None, b["lw"].data_ptr(), st)

    s = 4 if ndt == 0 else 2
    out = {"shape": name, "task": task, "image": shape, "rows": n, "n_total": ntot, "row0": row0, "du": em.du,
           "dv": em.dv, "net_dtype": "f32" if ndt == 0 else "bf16", "buffer_sets": nsets,
           "working_set_MB": round(nsets * n * (8 * em.du + s * em.D) / 1e6, 1)}
    for label, fn, nbytes in (("finish", finish, n * (8 * em.du + 8 + s * em.D)),
                              ("finish_proposal_only", prop_only, n * (8 * em.du + s * em.du)),
                              ("finish_weights_only", lw_only, n * (4 + s * em.dv)),
                              ("concat", concat, n * (4 * em.du + s * em.D))):
        if ONLY and label not in ONLY:
            continue
        us_ = time_kernel(fn, nsets, iters)
        out[label] = {"us": round(us_, 2), "algorithmic_bytes": nbytes, "GBps": round(nbytes / us_ / 1e3, 1),
                      "frac_of_8TBps": round(nbytes / us_ / 1e3 / 8000.0, 4)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="c3,c3_ef,c4_shard,c5_shard,c5_whole")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="f32,bf16")
    ap.add_argument("--sliced", default="0,1")
    ap.add_argument("--sets", type=int, default=0, help="buffer sets to cycle through (0: enough for > 512 MB)")
    ap.add_argument("--only", default="", help="comma list of kernels to time (finish, finish_proposal_only, ...)")
    a = ap.parse_args()
    ONLY.update(x for x in a.only.split(",") if x)
    _lib.build()
    for nm in a.shapes.split(","):
        for dt in a.dtype.split(","):
            for sl in a.sliced.split(","):
                task, shape, n, _ = SHAPES[nm]
                per_set = n * shape[0] * shape[1] * shape[2] * 8
                nsets = a.sets or max(1, min(16, -(-600_000_000 // per_set)))
                run(nm, torch.float32 if dt == "f32" else torch.bfloat16, sl == "1", a.iters, nsets)
