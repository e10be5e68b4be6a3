#!/usr/bin/env python
"""Host-side enqueue cost per call (us) of the op kinds that dominate the step's launch count."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = "cuda"
M = int(os.environ.get("M", 64))   # tiny by default: GPU time negligible, pure host path
x = torch.randn(M, 512, device=dev, dtype=torch.bfloat16)
w = torch.randn(256, 512, device=dev, dtype=torch.bfloat16)
b = torch.randn(256, device=dev, dtype=torch.bfloat16)
g = torch.ones(512, device=dev)
xc = torch.randn(1, 24, 16, 16, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
wc = torch.randn(24, 24, 3, 3, device=dev, dtype=torch.bfloat16)
nat = dgtd.ops._native.ops()


def cost(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:44s} host {host:7.1f} us/call")


with torch.no_grad():
    cost("torch.addmm (hipBLASLt)", lambda: torch.addmm(b, x, w.t()))
    cost("torch.mm", lambda: torch.mm(x, w.t()))
    cost("F.linear", lambda: F.linear(x, w, b))
    cost("dgtd.linear (C++ binding)", lambda: nat.linear(x, w, b, 1))
    cost("dgtd.layer_norm (C++ binding)", lambda: nat.layer_norm(x, g, g, 1e-6))
    cost("dgtd.ops.layer_norm (python binding)", lambda: dgtd.ops.layernorm._LayerNormFn.apply(x, g, g, 1e-6))
    cost("F.conv2d 3x3 24->24 channels_last (MIOpen)", lambda: F.conv2d(xc, wc, None, padding=1))
    cost("torch elementwise add", lambda: x + x)
    cost("F.gelu", lambda: F.gelu(x))
    for bm in (True, False):
        torch.backends.cudnn.benchmark = bm
        cost(f"F.conv2d 3x3 24->24 cudnn.benchmark={bm}", lambda: F.conv2d(xc, wc, None, padding=1))
    torch.backends.cudnn.benchmark = False
xcr = xc.clone().requires_grad_()
wcr = wc.clone().requires_grad_()


def conv_fb():
    y = F.conv2d(xcr, wcr, None, padding=1)
    y.backward(y)


for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    cost(f"F.conv2d fwd+bwd cudnn.benchmark={bm}", conv_fb, 100)
torch.backends.cudnn.benchmark = False
xr = x.clone().requires_grad_()
wr = w.clone().requires_grad_()
br = b.clone().requires_grad_()


def fb():
    y = nat.linear(xr, wr, br, 1)
    y.backward(y)


cost("dgtd.linear fwd+bwd (autograd)", fb, 100)


def fb2():
    y = F.linear(xr, wr, br)
    y.backward(y)


cost("F.linear fwd+bwd (autograd)", fb2, 100)
