#!/usr/bin/env python
"""16 separate 3x3 convs (prompt decoders) vs one batched / grouped conv: device time (fwd+bwd) and host enqueue time."""
import time
import torch
import torch.nn.functional as F

dev, dt = "cuda", torch.bfloat16
B, H = 8, 128
x = torch.randn(B, 24, H, H, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_()
w1 = [torch.randn(24, 24, 3, 3, device=dev, dtype=dt, requires_grad=True) for _ in range(16)]
w2 = [torch.randn(24, 24, 3, 3, device=dev, dtype=dt, requires_grad=True) for _ in range(16)]


def separate():
    hs = [F.relu(F.conv2d(F.relu(F.conv2d(x, a, None, padding=1)), b, None, padding=1)) for a, b in zip(w1, w2)]
    torch.autograd.backward(hs, [torch.ones_like(h) for h in hs])


def batched():
    W1 = torch.cat(w1, 0)                      # [384, 24, 3, 3]
    W2 = torch.cat(w2, 0)                      # [384, 24, 3, 3] with groups=16
    h = F.relu(F.conv2d(x, W1, None, padding=1))
    h = F.relu(F.conv2d(h, W2, None, padding=1, groups=16))
    h.backward(torch.ones_like(h))


for name, fn in (("separate 16x(conv,conv)", separate), ("batched conv + grouped conv", batched)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); a.record()
    for _ in range(10):
        fn()
    b.record(); host = (time.perf_counter() - t) / 10 * 1e3
    torch.cuda.synchronize()
    print(f"{name:32s} device+host wall {a.elapsed_time(b) / 10:7.2f} ms   host enqueue {host:6.2f} ms")
