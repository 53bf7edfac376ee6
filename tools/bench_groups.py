"""How should a batch of chains be split over handles / streams?  For BASELINE config 2 (N = 65 536, T = 500) and a batch
of `--chains` chains, time `nsweeps` chain sweeps for every way of cutting the batch into G groups, and split the wall
time into what the HOST spends enqueueing (the call returns when every graph launch has been queued) and the rest.

    python tools/bench_groups.py [--chains 4] [--nsweeps 6]

Re-run under GPU_MAX_HW_QUEUES=8 (read by the HIP runtime at start-up) to see whether the groups' streams share
hardware queues.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--nsweeps", type=int, default=6)
    ap.add_argument("--nparticles", type=int, default=65536)
    ap.add_argument("--nsteps", type=int, default=500)
    a = ap.parse_args()
    import fbs_amd
    from fbs_amd import _lib, ops
    from fbs_amd.sdes import StationaryConstLinearSDE
    dev = torch.device("cuda:0")
    N, T, Cn = a.nparticles, a.nsteps, a.chains
    ts = np.linspace(0.0, 2.0, T + 1)
    br = fbs_amd.LinearGaussianBridge(np.array([-1.0, 1.0]), np.array([[2.0, 0.4], [0.4, 0.5]]),
                                      StationaryConstLinearSDE(a=-0.5, b=1.0), ts, du=1, device=dev)
    y0 = torch.zeros(1, device=dev)
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}  chains={Cn}  N={N}  T={T}")
    for G in [g for g in (1, 2, 3, 4, 6, 8) if Cn % g == 0 and g <= Cn]:
        os.environ["FBSMI_CHAIN_GROUPS"] = str(G)
        sw = fbs_amd.linear_gaussian.LGSweep(br, N, True, False, False, Cn)
        kt = torch.from_numpy(np.asarray(ops.PRNGKey(5)).astype(np.uint32).view(np.int32).copy()).to(dev).reshape(1, 2)
        x0 = torch.zeros((Cn, 1), device=dev)
        bs = torch.zeros((Cn, T + 1), dtype=torch.int32, device=dev)

        def run(n):
            if sw.children:
                _lib.call("fbsmi_lg_gibbs_chain_groups", sw._harr, len(sw.children), kt.data_ptr(), x0.data_ptr(), y0.data_ptr(),
                          bs.data_ptr(), n, None, 1, ops._stream())
            else:
                _lib.call("fbsmi_lg_gibbs_chain", sw.h, kt.data_ptr(), x0.data_ptr(), y0.data_ptr(), bs.data_ptr(), n, None, 1,
                          ops._stream())
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.nsweeps)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per = (t2 - t0) / a.nsweeps
        print(f"groups={G} x {Cn // G} chain(s): {per * 1e3:7.3f} ms per sweep of the batch = {per / T * 1e6:6.2f} us per step = "
              f"{N * T * Cn / per / 1e9:6.2f} G particle-steps/s; host enqueue {(t1 - t0) / a.nsweeps * 1e3:6.3f} ms per sweep "
              f"({(t1 - t0) / a.nsweeps / G / (2 * T + 12) * 1e6:5.2f} us per graph node)", flush=True)
        del sw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
