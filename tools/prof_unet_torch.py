"""Which kernels the UNet spends its time in (torch.profiler, no rocprofv3 in the way): config 3 / 5 batch shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fbs_amd.unet import UNet
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
B, H, C = (1024, 28, 1) if name == "c3" else (512, 64, 3)
net = UNet(dt=0.01, dim=64, in_channels=C, upsampling="pixel_shuffle").to(dev).eval()
x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return net(x, 0.7)
for _ in range(3): fwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fwd()
e1.record(); e1.synchronize()
print(f"{name}: {e0.elapsed_time(e1)/5:.2f} ms per forward of {B} images")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): fwd()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=90))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
    for _ in range(3): fwd()
    torch.cuda.synchronize()
# self device time per aten op and input shapes, convolutions aside: the glue between the network's kernels
rows = [e for e in prof2.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in prof2.key_averages() if e.self_device_time_total > 0)
print(f"self device time by aten op (3 forwards, total {tot/1e3:.2f} ms)")
for e in rows[:40]:
    print(f"{e.key:32s} {e.self_device_time_total/1e3:8.3f} ms {100*e.self_device_time_total/tot:5.1f}% x{e.count:4d}  {str(e.input_shapes)[:110]}")
