"""Reference rates of plain streaming kernels on cold buffers of the config-5 shard size (100 MB), measured the way
tools/bench_em.py measures: back-to-back launches cycling through buffer sets larger than the Infinity Cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd import _lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, nsets, iters=10):
    for s in range(nsets): fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for s in range(nsets): fn(s)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (iters * nsets) * 1e3
for n, D in ((2048, 12288), (16384, 12288)):
    nsets = 6 if n == 2048 else 2
    a = [torch.randn(n, D, device=dev) for _ in range(nsets)]
    b = [torch.empty(n, D, device=dev) for _ in range(nsets)]
    idx = [torch.randperm(n, device=dev, dtype=torch.int32) for _ in range(nsets)]
    mb = n * D * 4 / 1e6
    t = timeit(lambda i: b[i].copy_(a[i]), nsets); print(f"{n}x{D} torch copy_: {t:.1f} us, {2*mb/t*1e-3:.2f} TB/s (read+write)")
    t = timeit(lambda i: b[i].fill_(1.0), nsets); print(f"{n}x{D} torch fill_: {t:.1f} us, {mb/t*1e-3:.2f} TB/s (write)")
    t = timeit(lambda i: torch.sum(a[i]), nsets); print(f"{n}x{D} torch sum: {t:.1f} us, {mb/t*1e-3:.2f} TB/s (read)")
    t = timeit(lambda i: _lib.call("fbsmi_gather_rows", a[i].data_ptr(), idx[i].data_ptr(), n, D, b[i].data_ptr(), st), nsets)
    print(f"{n}x{D} fbsmi_gather_rows: {t:.1f} us, {2*mb/t*1e-3:.2f} TB/s (read+write)")
    t = timeit(lambda i: torch.add(a[i], 1.0, out=b[i]), nsets); print(f"{n}x{D} torch add out=: {t:.1f} us, {2*mb/t*1e-3:.2f} TB/s (read+write)")
