from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native


def _launch_fwd(x, wt_ptr, bias_ptr, aux, mode, K):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    L.call("dgtd_dwconv_fwd", L.ptr(x), wt_ptr, bias_ptr, L.ptr(aux), L.ptr(y), B, H, W, C, K, mode,
           L.dtype_code(x), L.stream_ptr(), algo=("hbm", (3 if mode == 2 else 2) * x.element_size() * x.numel()),
           key=f"dgtd_dwconv_fwd[k{K},mode{mode},{H}x{W}x{C}]")
    return y


class _DwConvFn(Function):
    """Depthwise KxK conv (+bias, optionally + exact GELU) on NHWC [B,H,W,C]
    (convnext_Block.dwconv cod.py:1095; DWConv + GELU of Mlp cod.py:1523-1531, :854-855)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu):
        L.check_cuda(x, weight)
        C, K = weight.shape[0], weight.shape[-1]
        KK = K * K
        assert x.shape[-1] == C and weight.shape[1] == 1 and (bias is None or bias.dtype == weight.dtype)
        packed = torch.empty((2 * KK + 1) * C, dtype=torch.float32, device=x.device)   # { w_t | w_t flipped | bias }
        L.call("dgtd_dwconv_pack", L.ptr(weight), L.ptr(bias), L.ptr(packed), C, K, L.dtype_code(weight), L.stream_ptr())
        base = packed.data_ptr()
        y = _launch_fwd(x, base, base + 8 * KK * C if bias is not None else None, None, 1 if gelu else 0, K)
        ctx.save_for_backward(x, packed)
        ctx.meta = (gelu, K, bias is not None, weight.shape, weight.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, packed = ctx.saved_tensors
        gelu, K, has_bias, wshape, wdtype = ctx.meta
        B, H, W, C = x.shape
        KK = K * K
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        base = packed.data_ptr()
        bias_ptr = base + 8 * KK * C if has_bias else None
        du = _launch_fwd(x, base, bias_ptr, dy, 2, K) if gelu else dy      # through the GELU: recompute the pre-activation
        dx = _launch_fwd(du, base + 4 * KK * C, None, None, 0, K)          # bwd-data = same kernel, flipped filter
        grads = torch.empty((KK + 1) * C, dtype=torch.float32, device=x.device)   # { dw_t | db }
        ws = torch.empty(L.load().dgtd_dwconv_bwd_weight_workspace(B, H, W, C, K), dtype=torch.uint8, device=x.device)
        gb = grads.data_ptr()
        L.call("dgtd_dwconv_bwd_weight", L.ptr(x), L.ptr(du), gb, 1 if has_bias else 0, L.ptr(ws), B, H, W, C, K,
               L.dtype_code(x), L.stream_ptr(), algo=("hbm", 2 * x.element_size() * x.numel()),
               key=f"dgtd_dwconv_bwd_weight[k{K},{H}x{W}x{C}]")
        dw = torch.empty(wshape, dtype=wdtype, device=x.device)
        db = torch.empty(C, dtype=wdtype, device=x.device) if has_bias else None
        L.call("dgtd_dwconv_unpack_grads", gb, L.ptr(dw), L.ptr(db), C, K, L.dtype_code(dw), L.stream_ptr())
        return dx, dw, db, None


def dwconv_nhwc(x: torch.Tensor, weight: torch.Tensor, bias, gelu: bool = False) -> torch.Tensor:
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.dwconv_nhwc(x, weight, bias, bool(gelu))
    return _DwConvFn.apply(x.contiguous(), weight.contiguous(), bias, gelu)


def dwconv_fork(x, weight, bias=None):
    """(dwconv(x), x) for a residual block whose skip connection starts at the convolution's input (convnext_Block, cod.py:1104-1116):
    use the second value for the skip; with the C++ bindings the skip gradient is added inside the input-gradient convolution
    (dgtd_dwconv_fwd mode 3) instead of a separate elementwise add."""
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.dwconv_fork(x, weight, bias)
    return dwconv_nhwc(x, weight, bias), x
