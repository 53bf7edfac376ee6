"""explicit_final on a power-of-two ensemble (N + 1 = 2^k + 1 slots): per-step time of the fused sweep with the two-launch
step (default) and, with FBSMI_TREE_PLUS1=0, with the three-launch step it replaced.  python tools/bench_plus1.py [nparticles]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd  # noqa: E402
from fbs_amd.sdes import StationaryConstLinearSDE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T, C = 500, 4
dev = torch.device("cuda:0")
br = fbs_amd.LinearGaussianBridge(np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]), StationaryConstLinearSDE(a=-0.5, b=1.0),
                                  np.linspace(0.0, 2.0, T + 1), du=1, device=dev)
for ef in (True, False):
    sw = br.sweep_handle(n, True, ef, nchains=C)
    y0 = np.zeros(1, np.float32)
    k, x, b, _ = sw.chain(fbs_amd.PRNGKey(1), np.zeros((C, 1), np.float32), y0, np.zeros((C, T + 1), np.int32), 2, keep=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sw.chain(k, x, y0, b, 10, keep=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"nparticles={n} explicit_final={ef} ({n + int(ef)} slots), {C} chains, FBSMI_TREE_PLUS1={os.environ.get('FBSMI_TREE_PLUS1', '1')}: "
          f"{dt * 1e3:.3f} ms per sweep = {dt / T * 1e6:.2f} us per step")
