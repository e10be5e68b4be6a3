from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L

_MIX = {}


def _mix(weights, device):
    key = (tuple(float(w) for w in weights), device)
    t = _MIX.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=torch.float32, device=device)
        _MIX[key] = t
    return t


class _SegLossFn(Function):
    """sum_k mix[k] * cal_loss(bilinear_x8(lo[k]), label)  (twig/model/cod.py:76-85, :137-142, :796, :806) without ever
    materialising the five full-resolution logit maps.  lo: [5, B, hs, hs] fp32 low-resolution head outputs."""

    @staticmethod
    def forward(ctx, lo, label, mix):
        L.check_cuda(lo, label, mix)
        K, B, hs, _ = lo.shape
        S = label.shape[-1]
        assert K == 5 and label.numel() == B * S * S and lo.dtype == torch.float32 and label.dtype == torch.float32
        ws = torch.empty(L.load().dgtd_seg_loss_workspace(B, S), dtype=torch.uint8, device=lo.device)
        loss = torch.empty(1, dtype=torch.float32, device=lo.device)
        L.call("dgtd_seg_loss_fwd", L.ptr(lo), L.ptr(label), L.ptr(mix), L.ptr(loss), L.ptr(ws), B, S, hs, L.stream_ptr(),
               algo=("hbm", 4.0 * B * S * S * 4), key=f"dgtd_seg_loss_fwd[B={B},S={S}]")
        ctx.save_for_backward(lo, label, mix, ws)
        return loss[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lo, label, mix, ws = ctx.saved_tensors
        K, B, hs, _ = lo.shape
        S = label.shape[-1]
        g = g.reshape(1).float().contiguous()
        dlo = torch.empty_like(lo)
        L.call("dgtd_seg_loss_bwd", L.ptr(lo), L.ptr(label), L.ptr(mix), L.ptr(g), L.ptr(dlo), L.ptr(ws), B, S, hs, L.stream_ptr(),
               algo=("hbm", 4.0 * B * S * S * 2 * 4), key=f"dgtd_seg_loss_bwd[B={B},S={S}]")
        return dlo, None, None


def seg_loss(lowres_maps, label, weights=(0.0, 0.2, 0.4, 0.6, 1.0)):
    """lowres_maps: five [B,1,hs,hs] tensors (P1[0..3] and P2 before their x8 up-sampling)."""
    from .hitnet import stack
    lo = stack([m.float().squeeze(1) for m in lowres_maps])
    return _SegLossFn.apply(lo, label.float().contiguous(), _mix(weights, lo.device))


def ssim_value(x_hp: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
    """loss3 of cod.forward (cod.py:143-144): SSIM(minmax(x_hp), image) with the reference's SSIM module (cod.py:316-351), value only
    (it has no gradient path to any parameter).  fp32 [B,C,S,S] inputs, one fused pass (dgtd_ssim_value)."""
    x = x_hp.detach().float().contiguous()
    y = image.detach().float().contiguous()
    L.check_cuda(x, y)
    B, C, S, S2 = x.shape
    assert S == S2 and y.shape == x.shape
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(16, dtype=torch.uint8, device=x.device)
    L.call("dgtd_ssim_value", L.ptr(x), L.ptr(y), L.ptr(out), L.ptr(ws), B, C, S, L.stream_ptr())
    return out[0]
