"""The `spill` workload of bench.py alone (BASELINE config 2's model at N = 2^22 particles x 4 chains: ~400 MB of per-step
arrays, beyond the Infinity Cache), for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/spill_prof -o run -- python3 tools/spill_run.py
    rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/spill_run.py --sweeps 1

Prints particle-steps/s of the timed sweeps.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--nparticles", type=int, default=0)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--nsteps", type=int, default=20)
    ap.add_argument("--sweeps", type=int, default=3)
    a = ap.parse_args()
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    dev = torch.device("cuda:0")
    N, T, C = (a.nparticles or (1 << a.log2n)), a.nsteps, a.chains
    br = fbs_amd.LinearGaussianBridge(np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]),
                                      StationaryConstLinearSDE(a=-0.5, b=1.0), np.linspace(0.0, 2.0, T + 1), du=1, device=dev)
    sw = br.sweep_handle(N, True, False, nchains=C)
    y0 = np.zeros(1, np.float32)
    k, x, b, _ = sw.chain(fbs_amd.PRNGKey(1), np.zeros((C, 1), np.float32), y0, np.zeros((C, T + 1), np.int32), 1, keep=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    sw.chain(k, x, y0, b, a.sweeps, keep=False)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / a.sweeps
    print(f"N={N} chains={C} T={T}: {dt * 1e3:.3f} ms per sweep = {dt / T * 1e6:.1f} us per step = {N * T * C / dt / 1e9:.3f} G particle-steps/s "
          f"= {32.0 * N * T * C / dt / 1e9:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
