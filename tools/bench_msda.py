#!/usr/bin/env python
"""Device time of the dgtd multi-scale deformable attention kernels at a Deformable-DETR encoder shape, next to the PyTorch
composition the reference keeps "for debug and test only" (ms_deform_attn_core_pytorch: per-level F.grid_sample)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = "cuda"
N, M, D, P = 8, 8, 32, 4
shapes = [(64, 64), (32, 32), (16, 16), (8, 8)]
L, S = len(shapes), sum(h * w for h, w in shapes)
Lq = S
shp = torch.tensor(shapes, device=dev)
lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
value = (torch.rand(N, S, M, D, device=dev) * 0.01).requires_grad_()
loc = torch.rand(N, Lq, M, L, P, 2, device=dev).requires_grad_()
attn = torch.rand(N, Lq, M, L, P, device=dev)
attn = (attn / attn.sum((-1, -2), keepdim=True)).requires_grad_()
g = torch.randn(N, Lq, M * D, device=dev)


def composite(value, loc, attn):
    levels = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * loc - 1
    out = []
    for lid, (h, w) in enumerate(shapes):
        v = levels[lid].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        gr = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        out.append(F.grid_sample(v, gr, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = attn.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    return (torch.stack(out, dim=-2).flatten(-2) * a).sum(-1).view(N, M * D, Lq).transpose(1, 2).contiguous()


def timeit(fn, n=10):
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


def ours_f():
    return dgtd.ops.ms_deform_attn(value, shp, lsi, loc, attn, 64)


def ours_fb():
    torch.autograd.grad(ours_f(), (value, loc, attn), g)


def comp_fb():
    torch.autograd.grad(composite(value, loc, attn), (value, loc, attn), g)


err = (ours_f() - composite(value, loc, attn)).abs().max().item()
tf, tfb, cf, cfb = timeit(ours_f), timeit(ours_fb), timeit(lambda: composite(value, loc, attn)), timeit(comp_fb)
alg = 4 * (N * Lq * M * D * (4 * L * P + 1))
print(f"MSDeformAttn N={N} Lq={Lq} M={M} D={D} L={L} P={P} fp32: max|dgtd - composite| = {err:.2e}")
print(f"  dgtd fwd {tf:8.1f} us ({alg / tf / 1e3:.0f} GB/s of gathered bytes)   fwd+bwd {tfb:8.1f} us")
print(f"  torch composition (grid_sample per level) fwd {cf:8.1f} us   fwd+bwd {cfb:8.1f} us   -> {cf / tf:.1f}x / {cfb / tfb:.1f}x")
