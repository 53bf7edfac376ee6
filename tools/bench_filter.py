"""Per-step time of the fused bootstrap_filter / pmcmc_filter_step of the 2-D toy at N = 65 536 (a power of two: the
two-launch step with tree-walking searches; FBSMI_TREE_STEP=0 keeps the cdf launch) and at N = 70 000 (three launches).
python tools/bench_filter.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fbs_amd
from fbs_amd.sdes import StationaryConstLinearSDE
dev = torch.device("cuda:0")
T = 200
ts = np.linspace(0.0, 2.0, T + 1)
br = fbs_amd.LinearGaussianBridge(np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]), StationaryConstLinearSDE(a=-0.5, b=1.0), ts, du=1, device=dev)
for N in (4096, 65536, 70000):
    for flow in ("bootstrap", "pmcmc"):
        h = br.filter_handle(N, flow, "stratified")
        vs = torch.zeros((T + 1, 1), device=dev)
        init = torch.randn((N, 1), device=dev)
        key = fbs_amd.PRNGKey(3)
        for _ in range(2): h.run(key, vs, init)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n): h.run(key, vs, init)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"N={N} {flow}: {dt * 1e3:.3f} ms per filter run = {dt / T * 1e6:.2f} us per step (FBSMI_TREE_STEP={os.environ.get('FBSMI_TREE_STEP', '1')})")
