from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


def _launch_fwd(x, wt, bias, aux, mode, K):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    L.call("dgtd_dwconv_fwd", L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(aux), L.ptr(y), B, H, W, C, K, mode,
           L.dtype_code(x), L.stream_ptr(), algo=("hbm", (3 if mode == 2 else 2) * x.element_size() * x.numel()),
           key=f"dgtd_dwconv_fwd[k{K},mode{mode},{H}x{W}x{C}]")
    return y


class _DwConvFn(Function):
    """Depthwise KxK conv (+bias, optionally + exact GELU) on NHWC [B,H,W,C]
    (convnext_Block.dwconv cod.py:1095; DWConv + GELU of Mlp cod.py:1523-1531, :854-855)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu):
        L.check_cuda(x)
        C, K = weight.shape[0], weight.shape[-1]
        assert x.shape[-1] == C and weight.shape[1] == 1
        wt = weight.detach().float().reshape(C, K * K).t().contiguous()   # [K*K, C]: lanes along C read contiguous taps
        b32 = bias.detach().float().contiguous() if bias is not None else None
        y = _launch_fwd(x, wt, b32, None, 1 if gelu else 0, K)
        ctx.save_for_backward(x, wt, b32 if b32 is not None else torch.empty(0, device=x.device))
        ctx.meta = (gelu, K, bias is not None, weight.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wt, b32 = ctx.saved_tensors
        gelu, K, has_bias, wshape = ctx.meta
        B, H, W, C = x.shape
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        bias = b32 if has_bias else None
        du = _launch_fwd(x, wt, bias, dy, 2, K) if gelu else dy           # through the GELU: recompute the pre-activation
        dx = _launch_fwd(du, wt.flip(0).contiguous(), None, None, 0, K)    # bwd-data = same kernel, flipped filter
        dwt = torch.zeros_like(wt)
        db = torch.zeros(C, dtype=torch.float32, device=x.device) if has_bias else None
        L.call("dgtd_dwconv_bwd_weight", L.ptr(x), L.ptr(du), L.ptr(dwt), L.ptr(db), B, H, W, C, K, L.dtype_code(x),
               L.stream_ptr(), algo=("hbm", 2 * x.element_size() * x.numel()), key=f"dgtd_dwconv_bwd_weight[k{K},{H}x{W}x{C}]")
        return dx, dwt.t().reshape(wshape), db, None


def dwconv_nhwc(x: torch.Tensor, weight: torch.Tensor, bias, gelu: bool = False) -> torch.Tensor:
    return _DwConvFn.apply(x.contiguous(), weight, bias, gelu)
