"""Dense 3x3 convolution (stride 1, padding 1) on NHWC bf16 / fp16 maps with 24..96 channels over the C ABI (csrc/conv3x3.hip):
the conv+ReLU pairs of the prompt decoders (twig/model/cod.py:1216-1226) and the CAB bodies (cod.py:441-446).
Z independent convolutions run in one launch."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native


def supported(x: torch.Tensor, Ci: int, Co: int, H: int, W: int) -> bool:
    return bool(x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and L.load().dgtd_conv3x3_supported(Ci, Co, H, W))


class _Conv3x3Fn(Function):
    """x [Z|1,B,H,W,Ci], w [Z,Co,3,3,Ci], b [Z,Co] or None -> y [Z,B,H,W,Co] (optionally ReLU'd)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        L.check_cuda(x, w)
        Z, Co, _, _, Ci = w.shape
        shared = x.shape[0] == 1 and Z > 1
        _, B, H, W, _ = x.shape
        y = torch.empty(Z, B, H, W, Co, dtype=x.dtype, device=x.device)
        bc = b.contiguous() if b is not None else None
        flops = 2.0 * Z * B * H * W * 9 * Ci * Co
        L.call("dgtd_conv3x3_fwd", L.ptr(x), None, L.ptr(w), L.ptr(bc), L.ptr(y), Z, B, H, W, Ci, Co, int(relu), int(shared),
               L.dtype_code(x), L.stream_ptr(), algo=("hbm", 2 * (x.numel() + y.numel())), key=f"dgtd_conv3x3_fwd[Z={Z},{H}x{W},{Ci}->{Co}]")
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.meta = (relu, shared, b is not None, flops)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu, shared, has_b, _ = ctx.meta
        Z, Co, _, _, Ci = w.shape
        _, B, H, W, _ = x.shape
        dy = (dy if dy.dtype == x.dtype else dy.to(x.dtype)).contiguous()
        st = L.stream_ptr()
        dx = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(Z, Ci, 3, 3, Co, dtype=w.dtype, device=w.device)
            L.call("dgtd_conv3x3_flip", L.ptr(w), L.ptr(wt), Z, Co, Ci, st)
            dx = torch.empty(Z, B, H, W, Ci, dtype=x.dtype, device=x.device)
            L.call("dgtd_conv3x3_fwd", L.ptr(dy), L.ptr(y), L.ptr(wt), None, L.ptr(dx), Z, B, H, W, Co, Ci, 0, 0, L.dtype_code(dy), st,
                   algo=("hbm", 2 * (dx.numel() + (2 if relu else 1) * dy.numel())), key=f"dgtd_conv3x3_bwd_x[Z={Z},{H}x{W},{Co}->{Ci}]")
            if shared:
                dx = dx.sum(0, keepdim=True)
        dw = torch.empty_like(w)
        db = torch.empty(Z, Co, dtype=w.dtype, device=w.device) if has_b else None
        ws = torch.empty(L.load().dgtd_conv3x3_wgrad_workspace(Z, B, H, W, Ci, Co), dtype=torch.uint8, device=x.device)
        L.call("dgtd_conv3x3_wgrad", L.ptr(x), L.ptr(dy), L.ptr(y), L.ptr(dw), L.ptr(db), L.ptr(ws), Z, B, H, W, Ci, Co, int(shared), L.dtype_code(x), st,
               algo=("hbm", 2 * ((1 if shared else Z) * B * H * W * Ci + (2 if relu else 1) * dy.numel())),
               key=f"dgtd_conv3x3_wgrad[Z={Z},{H}x{W},{Ci}->{Co}]")
        return dx, dw, db, None


def _ohwi(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [O,I,3,3] -> [O,3,3,I]: a view when the parameter is stored channels_last (dist.GradReducer does that)."""
    return w.permute(0, 2, 3, 1).contiguous()


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """F.conv2d(x, weight, bias, padding=1) (+ReLU) for a logical [B,C,H,W] tensor; the result is a channels_last view."""
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.conv3x3_cl(x, weight, bias, relu)
    xn = x.permute(0, 2, 3, 1).contiguous().unsqueeze(0)               # NHWC: a view for channels_last inputs
    y = _Conv3x3Fn.apply(xn, _ohwi(weight).unsqueeze(0), bias.unsqueeze(0) if bias is not None else None, relu)
    return y[0].permute(0, 3, 1, 2)


def conv3x3_stack(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Optional[Sequence[torch.Tensor]], relu: bool) -> torch.Tensor:
    """Z convolutions in one launch.  x: NHWC [B,H,W,Ci] shared by all, or [Z,B,H,W,Ci]; weights: Z Conv2d weights [Co,Ci,3,3].
    Returns NHWC [Z,B,H,W,Co]."""
    from .hitnet import stack
    w = stack([_w.permute(0, 2, 3, 1) for _w in weights])        # views for the O,H,W,I storage of the reducer's working copies
    b = stack(list(biases)) if biases is not None else None
    if x.ndim == 4:
        x = x.unsqueeze(0)
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.conv3x3(x, w, b, relu)
    return _Conv3x3Fn.apply(x.contiguous(), w, b, relu)
