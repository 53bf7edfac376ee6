"""Closure-tier primitives at the shapes of BASELINE configs 3-5 (per-GPU
N x du of the image experiments): time per call and algorithmic GB/s against the 8 TB/s HBM peak.

    python tools/bench_prims.py            # on a GPU box
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd  # noqa: E402
from fbs_amd import ops  # noqa: E402
from fbs_amd.samplers.csmc.resamplings import killing  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps   # us


def main():
    dev = torch.device("cuda", 0)
    key = fbs_amd.PRNGKey(1)
    rows = []
    shapes = [("C3 MNIST inpaint-15", 4096, 225), ("C4 MNIST SB supr-4 (per GPU)", 2048, 735),
              ("C5 CelebA-64 inpaint-32 (per GPU)", 2048, 3072), ("C5 whole ensemble", 16384, 3072)]
    for name, N, du in shapes:
        us = torch.randn(N, du, device=dev)
        lw = torch.randn(N, device=dev)
        w = ops.normalise(lw)
        idx = killing(key, w, 3, 5, True)
        t_g = timeit(lambda: ops.take_rows(us, idx))
        t_k = timeit(lambda: killing(key, w, 3, 5, True))
        t_n = timeit(lambda: ops.normalise(lw, log_space=True))
        t_r = timeit(lambda: ops.normal(key, (N, du), device=dev))
        t_s = timeit(lambda: ops.set_row(us, 7, us[0]))
        gb = 8.0 * N * du / 1e3   # read + write, KB -> GB/s with us
        rows.append({"config": name, "N": N, "du": du,
                     "gather_rows_us": t_g, "gather_rows_GBps": gb / t_g * 1e-3 * 1e3, "gather_frac_hbm": gb / t_g / PEAK,
                     "cond_killing_us": t_k, "normalise_us": t_n,
                     "normal_draw_us": t_r, "normal_draw_GBps": 4.0 * N * du / 1e3 / t_r, "set_row_us": t_s})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
