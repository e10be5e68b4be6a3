from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L

LAT, P = 24, 144


class _DiffuserFn(Function):
    """x_hp [B,3,S,S], depth [B,1,S,S] + the regressor / depth-embedding parameters -> propagated state
    x4 [B,24,12,12] (twig/model/cod.py:1295-1298 + MessagePassing steps cod.py:1193-1205)."""

    @staticmethod
    def forward(ctx, x_hp, depth, reg_w, reg_b, enc_w, enc_b):
        x_hp, depth = x_hp.float().contiguous(), depth.float().contiguous()
        rw, rb = reg_w.float().reshape(-1, 3).contiguous(), reg_b.float().contiguous()
        ew, eb = enc_w.float().reshape(-1).contiguous(), enc_b.float().contiguous()
        L.check_cuda(x_hp, depth, rw, rb, ew, eb)
        B, _, S, _ = x_hp.shape
        x4 = torch.empty(B, LAT, 12, 12, dtype=torch.float32, device=x_hp.device)
        L.call("dgtd_diffuser_fwd", L.ptr(x_hp), L.ptr(depth), L.ptr(rw), L.ptr(rb), L.ptr(ew), L.ptr(eb), L.ptr(x4), B, S,
               L.stream_ptr(), algo=("hbm", 4.0 * B * (4 * P + LAT * P)))
        ctx.save_for_backward(x_hp, depth, rw, rb, ew, eb)
        ctx.shapes = (reg_w.shape, enc_w.shape)
        return x4

    @staticmethod
    @once_differentiable
    def backward(ctx, g4):
        x_hp, depth, rw, rb, ew, eb = ctx.saved_tensors
        B, _, S, _ = x_hp.shape
        g4 = g4.float().contiguous()
        d_rw, d_rb, d_ew, d_eb = (torch.zeros_like(t) for t in (rw, rb, ew, eb))
        L.call("dgtd_diffuser_bwd", L.ptr(x_hp), L.ptr(depth), L.ptr(rw), L.ptr(rb), L.ptr(ew), L.ptr(eb), L.ptr(g4),
               L.ptr(d_rw), L.ptr(d_rb), L.ptr(d_ew), L.ptr(d_eb), B, S, L.stream_ptr())
        return None, None, d_rw.view(ctx.shapes[0]), d_rb, d_ew.view(ctx.shapes[1]), d_eb


class _DiffuseTailFn(Function):
    """fused = bilinear_{12->S}(conv1x1_{24->3}(x4)) + image (cod.py:1206-1207, :1302)."""

    @staticmethod
    def forward(ctx, x4, cw, cb, image):
        x4, image = x4.float().contiguous(), image.float().contiguous()
        w, b = cw.float().reshape(3, LAT).contiguous(), cb.float().contiguous()
        L.check_cuda(x4, image, w, b)
        B, _, S, _ = image.shape
        out = torch.empty_like(image)
        L.call("dgtd_diffuse_tail_fwd", L.ptr(x4), L.ptr(w), L.ptr(b), L.ptr(image), L.ptr(out), B, S, L.stream_ptr(),
               algo=("hbm", 2.0 * 4 * B * 3 * S * S))
        ctx.save_for_backward(x4, w)
        ctx.meta = (B, S, cw.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x4, w = ctx.saved_tensors
        B, S, wshape = ctx.meta
        gout = gout.float().contiguous()
        g4 = torch.empty_like(x4)
        d_w = torch.zeros_like(w)
        d_b = torch.zeros(3, dtype=torch.float32, device=x4.device)
        ws = torch.empty(L.load().dgtd_diffuse_tail_bwd_workspace(B), dtype=torch.uint8, device=x4.device)
        L.call("dgtd_diffuse_tail_bwd", L.ptr(gout), L.ptr(x4), L.ptr(w), L.ptr(g4), L.ptr(d_w), L.ptr(d_b), L.ptr(ws), B, S,
               L.stream_ptr(), algo=("hbm", 4.0 * B * 3 * S * S))
        return g4, d_w.view(wshape), d_b, None


def diffuser_state(x_hp, depth, reg_w, reg_b, enc_w, enc_b):
    return _DiffuserFn.apply(x_hp, depth, reg_w, reg_b, enc_w, enc_b)


def diffuse_tail(x4, conv_w, conv_b, image):
    return _DiffuseTailFn.apply(x4, conv_w, conv_b, image)
