from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native


class _LayerNormFn(Function):
    """LayerNorm over the last dim (nn.LayerNorm of twig/model/cod.py:979,881,929,936,1043)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        L.check_cuda(x, weight, bias)
        C = x.shape[-1]
        rows = x.numel() // C
        w32, b32 = weight.float(), bias.float()
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.call("dgtd_layernorm_fwd", L.ptr(x), L.ptr(w32), L.ptr(b32), L.ptr(y), L.ptr(mean), L.ptr(rstd),
               rows, C, float(eps), L.dtype_code(x), L.stream_ptr(),
               algo=("hbm", 2 * x.element_size() * rows * C), key=f"dgtd_layernorm_fwd[rows={rows},C={C}]")
        ctx.save_for_backward(x, w32, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w32, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.load().dgtd_layernorm_bwd_workspace(C), dtype=torch.uint8, device=x.device)
        L.call("dgtd_layernorm_bwd", L.ptr(dy), L.ptr(x), L.ptr(w32), L.ptr(mean), L.ptr(rstd), L.ptr(dx),
               L.ptr(dg), L.ptr(db), L.ptr(ws), rows, C, L.dtype_code(x), L.stream_ptr(),
               algo=("hbm", 3 * x.element_size() * rows * C), key=f"dgtd_layernorm_bwd[rows={rows},C={C}]")
        return dx, dg, db, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.layer_norm(x, weight, bias, eps)
    return _LayerNormFn.apply(x.contiguous(), weight, bias, eps)


def layer_norm_fork(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float):
    """(LayerNorm(x), x) for pre-norm residual blocks: use the second value for the skip connection.  With the C++ bindings both
    gradients arrive at one node and are added inside the LayerNorm backward kernel (dgtd_layernorm_bwd_add)."""
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.layer_norm_fork(x, weight, bias, eps)
    return _LayerNormFn.apply(x.contiguous(), weight, bias, eps), x
