"""fbsmi_nn_conv3x3 against MIOpen's convolution (bf16, channels_last) at the network's layer shapes.
python tools/bench_conv.py [c3|c5]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd.unet import _conv3x3_hip
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
B, H0 = (1024, 28) if which == "c3" else (512, 64)
only = int(sys.argv[2]) if len(sys.argv) > 2 else None
shapes = [(H0, 64, 64), (H0, 128, 64), (H0 // 2, 64, 64), (H0 // 2, 128, 128), (H0 // 2, 128, 512), (H0 // 4, 128, 128), (H0 // 4, 128, 256)]
if only is not None:
    shapes = shapes[:only]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, cin, cout in shapes:
    x = torch.randn(B, cin, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_mi = timeit(lambda: torch.nn.functional.conv2d(x, w, None, padding=1))
    t_my = timeit(lambda: _conv3x3_hip(x, w, None))
    fl = 2.0 * B * H * H * cin * cout * 9
    by = 2.0 * B * H * H * (cin + cout)
    print(f"{B}x{H}x{H} {cin:4d}->{cout:4d}: MIOpen {t_mi:7.1f} us   libfbsmi {t_my:7.1f} us  ({fl / t_my / 1e9:.2f} PFLOP/s, {by / t_my / 1e6:.2f} TB/s)")
