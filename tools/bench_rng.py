import sys, os, torch
sys.path.insert(0, os.getcwd())
from fbs_amd import _lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for n in (2048*3072, 16384*3072):
    outs = [torch.empty(n, device=dev) for _ in range(4)]
    for mode, name in ((0, "fbsmi_random_bits"), (1, "fbsmi_uniform"), (2, "fbsmi_normal")):
        fn = lambda i: _lib.call(name, 1, 2, n, outs[i % 4].data_ptr(), st)
        for i in range(4): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): fn(i)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"{name} n={n}: {us:.2f} us  {n/us/1e3:.1f} G elements/s  {4*n/us/1e3:.0f} GB/s written")
