#!/usr/bin/env python
"""Run ONE dgtd op at its config-2 shape a few times, for the rocprofv3 --pmc passes that feed bench.py's `roofline.traffic`:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/<op>/fetch -- python3 tools/pmc_ops.py <op>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/<op>/write -- python3 tools/pmc_ops.py <op>
then tools/pmc_collect.py out > profiles/rNN_pmc_traffic.json"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NATIVE_OPS = ("linear_gelu_8192x512x2048", "linear_residual_8192x2048x512", "dwconv_batched_k7_n27_32x32x512", "mlp_residual_8192x512")   # nodes that only exist in the C++ bindings
os.environ.setdefault("DGTD_TORCH_BINDINGS", "1" if len(sys.argv) > 1 and sys.argv[1] in NATIVE_OPS else "0")
import dgtd  # noqa: E402

dev, bf = "cuda", torch.bfloat16
REPS = int(os.environ.get("REPS", 3))


def dw(K, H, C, gelu):
    x = torch.randn(8, H, H, C, device=dev, dtype=bf).requires_grad_()
    w = (torch.randn(C, 1, K, K, device=dev, dtype=bf) / K).requires_grad_()
    b = torch.randn(C, device=dev, dtype=bf).requires_grad_()
    dy = torch.randn(8, H, H, C, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.dwconv_nhwc(x, w, b, gelu)
        torch.autograd.grad(y, (x, w, b), dy)


def ln(rows, C):
    x = torch.randn(rows, C, device=dev, dtype=bf).requires_grad_()
    w = torch.ones(C, device=dev).requires_grad_()
    b = torch.zeros(C, device=dev).requires_grad_()
    dy = torch.randn(rows, C, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.layer_norm(x, w, b, 1e-6)
        torch.autograd.grad(y, (x, w, b), dy)


def colsum(rows, C):
    x = torch.randn(rows, C, device=dev, dtype=bf)
    for _ in range(REPS):
        dgtd.ops.colsum(x, bf)


def conv(Z, C, H, shared, relu):
    x = torch.randn(1 if shared else Z, 8, H, H, C, device=dev, dtype=bf).requires_grad_()
    ws = [torch.randn(C, C, 3, 3, device=dev, dtype=bf).mul_(0.05).contiguous(memory_format=torch.channels_last).requires_grad_() for _ in range(Z)]
    bs = [torch.randn(C, device=dev, dtype=bf).requires_grad_() for _ in range(Z)] if relu else None
    g = torch.randn(Z, 8, H, H, C, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.conv3x3_stack(x, ws, bs, relu)
        torch.autograd.grad(y, [x] + ws + (bs or []), g)


def attn():
    q = torch.randn(8, 16384, 64, device=dev, dtype=bf).requires_grad_()
    kv = torch.randn(8, 256, 128, device=dev, dtype=bf).requires_grad_()
    g = torch.randn(8, 16384, 64, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.sra_attention(q, kv, 1, 0.125)
        torch.autograd.grad(y, (q, kv), g)


def lin_gelu(rows, K, N):
    x = torch.randn(rows, K, device=dev, dtype=bf).requires_grad_()
    w = (torch.randn(N, K, device=dev, dtype=bf) / K ** 0.5).requires_grad_()
    b = torch.randn(N, device=dev, dtype=bf).requires_grad_()
    g = torch.randn(rows, N, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.linear_gelu(x, w, b)
        torch.autograd.grad(y, (x, w, b), g)


def lin_res(rows, K, N):
    h = torch.randn(rows, K, device=dev, dtype=bf).requires_grad_()
    w = (torch.randn(N, K, device=dev, dtype=bf) / K ** 0.5).requires_grad_()
    b = torch.randn(N, device=dev, dtype=bf).requires_grad_()
    x = torch.randn(8, rows // 8, N, device=dev, dtype=bf).requires_grad_()
    gamma = torch.ones(N, device=dev).requires_grad_()
    s = torch.ones(8, device=dev)
    g = torch.randn(8, rows // 8, N, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.linear_residual(h.view(8, rows // 8, K), w, b, x, s, gamma)
        torch.autograd.grad(y, (h, w, b, x, gamma), g)


def dw_batched(K, n, H, C):
    nat = dgtd.ops._native.ops()
    layers = [(torch.randn(8, H, H, C, device=dev, dtype=bf), (torch.randn(C, 1, K, K, device=dev, dtype=bf) / K).requires_grad_(),
               torch.randn(C, device=dev, dtype=bf).requires_grad_(), torch.randn(8, H, H, C, device=dev, dtype=bf)) for _ in range(n)]
    for _ in range(REPS):
        nat.set_deferred(True)
        for x, w, b, g in layers:
            torch.autograd.grad(dgtd.ops.dwconv_nhwc(x, w, b, False), (w, b), g)
        nat.set_deferred(False)


def mlp_res(rows, C):
    """convnext_Block's pointwise half as the ONE fused node on the package's own GEMM (csrc/gemm.hip): gemm_bias_gelu, gemm_bias_residual,
    scale_residual_bias_bwd, gemm_gelu_bwd, gemm_bias (input gradient) + the library weight-gradient GEMMs."""
    v = torch.randn(8, rows // 8, C, device=dev, dtype=bf).requires_grad_()
    x = torch.randn(8, rows // 8, C, device=dev, dtype=bf).requires_grad_()
    w1 = (torch.randn(4 * C, C, device=dev, dtype=bf) / C ** 0.5).requires_grad_()
    b1 = torch.randn(4 * C, device=dev, dtype=bf).requires_grad_()
    w2 = (torch.randn(C, 4 * C, device=dev, dtype=bf) / (4 * C) ** 0.5).requires_grad_()
    b2 = torch.randn(C, device=dev, dtype=bf).requires_grad_()
    gamma = torch.ones(C, device=dev).requires_grad_()
    s = torch.ones(8, device=dev)
    g = torch.randn(8, rows // 8, C, device=dev, dtype=bf)
    for _ in range(REPS):
        y = dgtd.ops.mlp_residual(v, w1, b1, w2, b2, x, s, gamma)
        torch.autograd.grad(y, (v, x, w1, b1, w2, b2, gamma), g)


OPS = {
    "mlp_residual_8192x512": lambda: mlp_res(8192, 512),
    "linear_gelu_8192x512x2048": lambda: lin_gelu(8192, 512, 2048),
    "linear_residual_8192x2048x512": lambda: lin_res(8192, 2048, 512),
    "dwconv_batched_k7_n27_32x32x512": lambda: dw_batched(7, 27, 32, 512),
    "dwconv_k7_32x32x512": lambda: dw(7, 32, 512, False),
    "dwconv_k7_128x128x128": lambda: dw(7, 128, 128, False),
    "dwconv_k3_128x128x512": lambda: dw(3, 128, 512, True),
    "layernorm_8192x512": lambda: ln(8192, 512),
    "colsum_8192x2048": lambda: colsum(8192, 2048),
    "conv3x3_96_64x64": lambda: conv(1, 96, 64, False, False),
    "conv3x3_Z16_24_128x128": lambda: conv(16, 24, 128, True, True),
    "attention_stage1": attn,
}
if __name__ == "__main__":
    OPS[sys.argv[1]]()
    torch.cuda.synchronize()
