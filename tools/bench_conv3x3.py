#!/usr/bin/env python
"""Device time of the dgtd conv3x3 kernels vs MIOpen (F.conv2d, channels_last bf16) at the shapes of the step."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (Z, B, C, H, shared, relu, bias) in [(16, 8, 24, 128, True, True, True), (16, 8, 24, 128, False, True, True), (1, 8, 24, 128, False, True, True),
                                          (1, 8, 96, 64, False, False, False), (1, 8, 64, 32, False, False, False), (1, 8, 32, 16, False, False, False),
                                          (1, 8, 64, 128, False, False, False)]:
    x = torch.randn(1 if shared else Z, B, H, H, C, device=dev, dtype=torch.bfloat16).requires_grad_()
    ws = [torch.randn(C, C, 3, 3, device=dev, dtype=torch.bfloat16).mul_(0.07).contiguous(memory_format=torch.channels_last).requires_grad_() for _ in range(Z)]
    bs = [torch.randn(C, device=dev, dtype=torch.bfloat16).requires_grad_() for _ in range(Z)] if bias else None
    g = torch.randn(Z, B, H, H, C, device=dev, dtype=torch.bfloat16)

    def ours_f():
        return dgtd.ops.conv3x3_stack(x, ws, bs, relu)

    def ours_fb():
        y = ours_f()
        torch.autograd.grad(y, [x] + ws + (bs or []), g)

    xn = [x[0 if shared else z].permute(0, 3, 1, 2) for z in range(Z)]
    gn = [g[z].permute(0, 3, 1, 2) for z in range(Z)]

    def lib_f():
        ys = []
        for z in range(Z):
            y = F.conv2d(xn[z], ws[z], bs[z] if bias else None, padding=1)
            ys.append(F.relu(y) if relu else y)
        return ys

    def lib_fb():
        ys = lib_f()
        torch.autograd.grad(ys, [x] + ws + (bs or []), gn)

    tf, tfb, lf, lfb = timeit(ours_f), timeit(ours_fb), timeit(lib_f), timeit(lib_fb)
    mb = 2 * 2 * Z * B * H * H * C / 1e6
    print(f"Z={Z:2d} B={B} {H}x{H} C={C:3d} shared={int(shared)} relu={int(relu)}: dgtd fwd {tf:8.1f} us  fwd+bwd {tfb:8.1f} us | "
          f"MIOpen fwd {lf:8.1f} us  fwd+bwd {lfb:8.1f} us   (fwd algorithmic {mb:.0f} MB)")
