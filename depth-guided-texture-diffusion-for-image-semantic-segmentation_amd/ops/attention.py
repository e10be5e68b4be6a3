from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native


class _SraAttnFn(Function):
    """softmax(Q K^T * scale) V with head_dim 64 (twig/model/cod.py:913-917), consuming the q / kv
    Linear outputs in place: q [B,N,h*64], kv [B,Nkv,2*h*64] -> out [B,N,h*64]."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        L.check_cuda(q, kv)
        B, N, C = q.shape
        Nkv = kv.shape[1]
        assert C == heads * 64 and kv.shape[2] == 2 * C and kv.dtype == q.dtype, (q.shape, kv.shape, heads)
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=q.device)
        L.call("dgtd_sra_attn_fwd", L.ptr(q), L.ptr(kv), L.ptr(out), L.ptr(lse), B, N, Nkv, heads, float(scale),
               L.dtype_code(q), L.stream_ptr(), algo=("mfma", 4.0 * B * heads * N * Nkv * 64),
               key=f"dgtd_sra_attn_fwd[N={N},Nkv={Nkv},h={heads}]")
        ctx.save_for_backward(q, kv, out, lse)
        ctx.heads, ctx.scale = heads, float(scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        B, N, C = q.shape
        Nkv = kv.shape[1]
        dout = dout.contiguous()
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        dq = torch.empty_like(q)
        dkv = torch.zeros(kv.shape, dtype=torch.float32, device=q.device)
        ws = torch.empty(L.load().dgtd_sra_attn_bwd_workspace(B, N, ctx.heads), dtype=torch.uint8, device=q.device)
        L.call("dgtd_sra_attn_bwd", L.ptr(q), L.ptr(kv), L.ptr(out), L.ptr(dout), L.ptr(lse), L.ptr(dq), L.ptr(dkv),
               L.ptr(ws), B, N, Nkv, ctx.heads, ctx.scale, L.dtype_code(q), L.stream_ptr(),
               algo=("mfma", 10.0 * B * ctx.heads * N * Nkv * 64), key=f"dgtd_sra_attn_bwd[N={N},Nkv={Nkv},h={ctx.heads}]")
        return dq, dkv.to(kv.dtype), None, None


def sra_attention(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    nat = _native.ops()
    if nat is not None and q.is_cuda:
        return nat.sra_attention(q, kv, heads, float(scale))
    return _SraAttnFn.apply(q.contiguous(), kv.contiguous(), heads, scale)
