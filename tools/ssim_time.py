import os, sys, torch
sys.path.insert(0, "/root/repo")
import dgtd
x = torch.rand(8, 3, 512, 512, device="cuda"); y = torch.randn(8, 3, 512, 512, device="cuda")
for _ in range(3): v = dgtd.ops.ssim_value(x, y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): v = dgtd.ops.ssim_value(x, y)
b.record(); torch.cuda.synchronize()
print("slices", os.environ.get("DGTD_SSIM_SLICES"), "us per call", round(a.elapsed_time(b) * 20, 1), float(v))
