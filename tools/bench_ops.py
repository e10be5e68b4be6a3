#!/usr/bin/env python
"""Per-kernel micro-benchmark at the shapes of BASELINE.json configs[1] (512x512, batch 8, bf16):
HIP-event timing of each C-ABI op, algorithmic bytes / flops -> achieved GB/s or TFLOP/s.

    python tools/bench_ops.py [--dtype bf16|f32] [--only dwconv,ln,attn,diffuser] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
only = set(filter(None, args.only.split(",")))
dev = "cuda"
B = 8


def timed(fn, iters=args.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def report(name, us, amount, kind):
    if kind == "hbm":
        print(f"{name:58s} {us:9.1f} us  {amount / us / 1e3:9.1f} GB/s   ({100 * amount / us / 1e3 / 8000:5.1f}% of 8 TB/s)")
    else:
        peak = 2500.0 if dt == torch.bfloat16 else 157.3
        print(f"{name:58s} {us:9.1f} us  {amount / us / 1e6:9.1f} TFLOP/s ({100 * amount / us / 1e6 / peak:5.1f}% of {peak:.0f})")


def want(k):
    return not only or k in only


e = 2 if dt == torch.bfloat16 else 4
if want("dwconv"):
    for K, H, C in [(7, 128, 128), (7, 64, 256), (7, 32, 512), (7, 16, 1024), (3, 128, 512), (3, 64, 1024), (3, 32, 1280), (3, 16, 2048)]:
        x = torch.randn(B, H, H, C, device=dev, dtype=dt)
        w = torch.randn(C, 1, K, K, device=dev) / K
        b = torch.randn(C, device=dev)
        dy = torch.randn_like(x)
        n = x.numel()
        xs = x.clone().requires_grad_()
        ws, bs = w.clone().requires_grad_(), b.clone().requires_grad_()
        gelu = K == 3
        report(f"dwconv_fwd k{K} {H}x{H}x{C}" + (" +gelu" if gelu else ""), timed(lambda: dgtd.ops.dwconv_nhwc(x, w, b, gelu)), 2 * e * n, "hbm")
        y = dgtd.ops.dwconv_nhwc(xs, ws, bs, gelu)
        report(f"dwconv_bwd (all) k{K} {H}x{H}x{C}", timed(lambda: torch.autograd.grad(y, (xs, ws, bs), dy, retain_graph=True)), (7 if gelu else 4) * e * n, "hbm")
if want("ln"):
    for rows, C in [(131072, 64), (32768, 128), (8192, 320), (2048, 512), (131072, 128), (32768, 256), (8192, 512), (2048, 1024)]:
        x = torch.randn(rows, C, device=dev, dtype=dt)
        w, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
        dy = torch.randn_like(x)
        xs = x.clone().requires_grad_()
        report(f"layernorm_fwd [{rows},{C}]", timed(lambda: dgtd.ops.layer_norm(x, w, b, 1e-6)), 2 * e * rows * C, "hbm")
        y = dgtd.ops.layer_norm(xs, w, b, 1e-6)
        report(f"layernorm_bwd [{rows},{C}]", timed(lambda: torch.autograd.grad(y, (xs, w, b), dy, retain_graph=True)), 3 * e * rows * C, "hbm")
if want("attn"):
    for N, Nkv, h in [(16384, 256, 1), (4096, 256, 2), (1024, 256, 5), (256, 256, 8)]:
        C = 64 * h
        q = torch.randn(B, N, C, device=dev, dtype=dt)
        kv = torch.randn(B, Nkv, 2 * C, device=dev, dtype=dt)
        do = torch.randn_like(q)
        qs, kvs = q.clone().requires_grad_(), kv.clone().requires_grad_()
        report(f"sra_attn_fwd N={N} Nkv={Nkv} h={h}", timed(lambda: dgtd.ops.sra_attention(q, kv, h, 0.125)), 4.0 * B * h * N * Nkv * 64, "mfma")
        o = dgtd.ops.sra_attention(qs, kvs, h, 0.125)
        report(f"sra_attn_bwd N={N} Nkv={Nkv} h={h}", timed(lambda: torch.autograd.grad(o, (qs, kvs), do, retain_graph=True)), 10.0 * B * h * N * Nkv * 64, "mfma")
if want("diffuser"):
    S = 512
    xhp = torch.rand(B, 3, S, S, device=dev)
    depth = torch.rand(B, 1, S, S, device=dev)
    img = torch.randn(B, 3, S, S, device=dev)
    rw, rb = torch.randn(1176, 3, 1, 1, device=dev), torch.randn(1176, device=dev)
    ew, eb = torch.randn(24, 1, 1, 1, device=dev), torch.randn(24, device=dev)
    cw, cb = torch.randn(3, 24, 1, 1, device=dev), torch.randn(3, device=dev)
    x4 = dgtd.ops.diffuser_state(xhp, depth, rw, rb, ew, eb)
    report("diffuser_state fwd (B=8, 24 ch, 12x12)", timed(lambda: dgtd.ops.diffuser_state(xhp, depth, rw, rb, ew, eb)), 4.0 * B * (4 * 144 + 24 * 144), "hbm")
    report("diffuse_tail fwd 512x512", timed(lambda: dgtd.ops.diffuse_tail(x4, cw, cb, img)), 2.0 * 4 * B * 3 * S * S, "hbm")
