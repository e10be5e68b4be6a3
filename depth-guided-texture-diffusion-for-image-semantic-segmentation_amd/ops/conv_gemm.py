"""Dense KxK convolutions as patch gather (csrc/im2col.hip) + library GEMM on token-major tensors: OverlapPatchEmbed.proj
(twig/model/cod.py:974-975, :1000), the folded prompt-decoder tails (cod.py:1220 + :1471), Hitnet.compress_out (cod.py:739), the
1-channel heads and - in fp32 parity mode - the 3x3 convolutions of the Hitnet decoder.  No MIOpen on the path: its fp32 Winograd
kernels cost gradient accuracy (see <package>/__init__.py) and every call pays layout transforms around the NHWC maps."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from .elementwise import linear


class _Im2ColFn(Function):
    """x logical [B,C,H,W] with ANY strides (NCHW image, channels_last map, offset view) -> col [B*Ho*Wo, K*K*C] in ``col_dtype``,
    column order (ky, kx, c).  Backward: col2im as a gather into a channels_last [B,C,H,W] gradient."""

    @staticmethod
    def forward(ctx, x, K, stride, pad, Ho, Wo, col_dtype):
        if not x.is_cuda:
            raise L.DgtdError("dgtd ops run only on the MI355X HIP device")
        B, C, H, W = x.shape
        sb, sc, sy, sx = x.stride()
        col = torch.empty(B * Ho * Wo, K * K * C, dtype=col_dtype, device=x.device)
        L.call("dgtd_im2col", L.ptr(x), L.ptr(col), B, H, W, C, sb, sy, sx, sc, K, stride, pad, Ho, Wo, L.dtype_code(x), L.dtype_code(col),
               L.stream_ptr())
        ctx.geom = (B, C, H, W, K, stride, pad, Ho, Wo, x.dtype)
        return col

    @staticmethod
    @once_differentiable
    def backward(ctx, dcol):
        B, C, H, W, K, stride, pad, Ho, Wo, xdtype = ctx.geom
        dcol = dcol.contiguous()
        dx = torch.empty((B, C, H, W), dtype=dcol.dtype, device=dcol.device, memory_format=torch.channels_last)
        L.call("dgtd_col2im", L.ptr(dcol), L.ptr(dx), B, H, W, C, K, stride, pad, Ho, Wo, L.dtype_code(dcol), L.stream_ptr())
        return (dx if dx.dtype == xdtype else dx.to(xdtype)), None, None, None, None, None, None


def conv2d_tokens(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, padding: int = 0) -> Tuple[torch.Tensor, int, int]:
    """F.conv2d(x, weight, bias, stride, padding) for dilation 1 / groups 1, returned token-major: ([B, Ho*Wo, O], Ho, Wo).
    ``weight`` [O,I,K,K]: its (ky,kx,ci)-ordered matrix is a free view when the parameter is stored O,H,W,I (dist.GradReducer)."""
    B, C, H, W = x.shape
    O, I, K, K2 = weight.shape
    assert I == C and K == K2, "conv2d_tokens: groups / non-square kernels are not on the path"
    Ho, Wo = (H + 2 * padding - K) // stride + 1, (W + 2 * padding - K) // stride + 1
    dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else weight.dtype
    if K == 1 and stride == 1 and padding == 0 and x.dtype == dt and x.permute(0, 2, 3, 1).is_contiguous():
        col = x.permute(0, 2, 3, 1).reshape(B * H * W, C)             # 1x1 conv on a channels_last map: the tokens themselves
    else:
        col = _Im2ColFn.apply(x, K, stride, padding, Ho, Wo, dt)
    wmat = weight.permute(0, 2, 3, 1).reshape(O, K * K * C)
    y = linear(col, wmat, bias)
    return y.view(B, Ho * Wo, O), Ho, Wo


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, padding: int = 0) -> torch.Tensor:
    """Same, as a logical [B,O,Ho,Wo] tensor in channels_last memory (a view of the token matrix)."""
    t, Ho, Wo = conv2d_tokens(x, weight, bias, stride, padding)
    return t.view(x.shape[0], Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
