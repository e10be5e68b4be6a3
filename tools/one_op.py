#!/usr/bin/env python
"""Run ONE op a few times (for rocprofv3 --pmc passes):  python tools/one_op.py dwconv_bww 7 128 128"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DGTD_TORCH_BINDINGS", "0")
import dgtd  # noqa: E402

op, K, H, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(8, H, H, C, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(C, 1, K, K, device="cuda") / K).requires_grad_()
b = torch.randn(C, device="cuda").requires_grad_()
dy = torch.randn_like(x)
xs = x.clone().requires_grad_()
for _ in range(3):
    y = dgtd.ops.dwconv_nhwc(xs, w, b, K == 3)
    torch.autograd.grad(y, (xs, w, b), dy)
torch.cuda.synchronize()
