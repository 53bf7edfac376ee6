"""Timing experiment: per-sweep time with only a subset of the four step kernels launched."""
import os, sys, time, subprocess, json
if len(sys.argv) > 1:
    if os.environ.get("PRELOAD_ROCM"):
        import ctypes
        for lib in ("libhsa-runtime64.so.1", "libamdhip64.so.7"):
            ctypes.CDLL("/opt/rocm/lib/" + lib, mode=ctypes.RTLD_GLOBAL)
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import fbs_amd
    from fbs_amd.sdes import StationaryConstLinearSDE
    N, T = int(os.environ.get("NP", 65536)), 500
    ts = np.linspace(0, 2, T + 1)
    br = fbs_amd.LinearGaussianBridge([-1., 1.], [[2., .4], [.4, .5]], StationaryConstLinearSDE(-0.5, 1.), ts, 1)
    sw = br.sweep_handle(N, True, False)
    key, x0, bs = fbs_amd.PRNGKey(1), np.zeros(1, np.float32), np.zeros(T + 1, np.int32)
    sw.chain(key, x0, [0.], bs, 2, keep=False); torch.cuda.synchronize()
    t = time.perf_counter(); sw.chain(key, x0, [0.], bs, 10, keep=False); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"mask {os.environ.get('FBSMI_DEBUG_STEP_MASK','15'):>2} N={N}: {dt*1e3:.3f} ms/sweep = {dt/T*1e6:.2f} us/step")
else:
    for pre in ("", "1"):
        for m in (0, 8, 15):
            print("preload /opt/rocm runtime:", bool(pre), flush=True)
            subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, FBSMI_DEBUG_STEP_MASK=str(m), PRELOAD_ROCM=pre))
