"""Chain-group splits of a batch of chains of BASELINE config 2 (FBSMI_CHAIN_GROUP_SIZES): python tools/bench_group_sizes.py C sizes [sizes ...]
e.g.  python tools/bench_group_sizes.py 16 8,8 6,5,5 4,4,4,4"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE  # noqa: E402

dev = torch.device("cuda:0")
N, T, C = 65536, 500, int(sys.argv[1])
br = fbs_amd.LinearGaussianBridge(np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]), StationaryConstLinearSDE(a=-0.5, b=1.0),
                                  np.linspace(0.0, 2.0, T + 1), du=1, device=dev)
for sizes in sys.argv[2:]:
    os.environ["FBSMI_CHAIN_GROUP_SIZES"] = sizes
    sw = fbs_amd.linear_gaussian.LGSweep(br, N, True, False, False, C)   # (not br.sweep_handle: that one caches per shape)
    y0 = np.zeros(1, np.float32)
    k, x, b, _ = sw.chain(fbs_amd.PRNGKey(1), np.zeros((C, 1), np.float32), y0, np.zeros((C, T + 1), np.int32), 2, keep=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k, x, b, _ = sw.chain(k, x, y0, b, 6, keep=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f"{C} chains as {sizes}: {dt * 1e3:.3f} ms per sweep = {dt / T * 1e6:.2f} us per step = {N * T * C / dt / 1e9:.2f} G particle-steps/s", flush=True)
    del sw
    torch.cuda.empty_cache()
