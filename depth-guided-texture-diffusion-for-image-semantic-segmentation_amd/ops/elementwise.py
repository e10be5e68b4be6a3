from __future__ import annotations

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native

_WS = {}
_CODES = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}   # compute dtypes of the C++ Linear nodes
SPLITK_WGRAD = os.environ.get("DGTD_SPLITK_WGRAD", "1") != "0"   # A/B switch for tools/ and bench runs


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Persistent per-device scratch for the two-stage column sums (stream-ordered reuse on the compute stream)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def colsum(x2d: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Column sums of a [rows, C] matrix (bias gradients), accumulated in fp32 and written as ``out_dtype``."""
    L.check_cuda(x2d)
    rows, C = x2d.shape
    if C % (16 // x2d.element_size()):   # narrower than one 16-byte chunk per lane (e.g. a 1-wide head): generic device reduction
        return x2d.sum(0, dtype=torch.float32).to(out_dtype)
    out = torch.empty(C, dtype=out_dtype, device=x2d.device)
    ws = _workspace(L.load().dgtd_colsum_workspace(C), x2d.device)
    L.call("dgtd_colsum", L.ptr(x2d), L.ptr(out), L.dtype_code(out), L.ptr(ws), rows, C, L.dtype_code(x2d), L.stream_ptr(),
           algo=("hbm", x2d.element_size() * x2d.numel()), key=f"dgtd_colsum[rows={rows},C={C}]")
    return out


class _ScaleResidualFn(Function):
    """out = x + s[b] * gamma[c] * y on [B, ..., C] (cod.py:1112-1116, :958-959); s = per-sample DropPath scale or None."""

    @staticmethod
    def forward(ctx, x, y, s, gamma):
        L.check_cuda(x, y)
        B, C = x.shape[0], x.shape[-1]
        rows = x.numel() // C
        g32 = gamma.detach().float().contiguous() if gamma is not None else None
        out = torch.empty_like(x)
        L.call("dgtd_scale_residual_fwd", L.ptr(x), L.ptr(y), L.ptr(s), L.ptr(g32), L.ptr(out), rows, C, rows // B,
               L.dtype_code(x), L.stream_ptr(), algo=("hbm", 3 * x.element_size() * x.numel()),
               key=f"dgtd_scale_residual_fwd[rows={rows},C={C}]")
        ctx.save_for_backward(y, s if s is not None else torch.empty(0, device=x.device),
                              g32 if g32 is not None else torch.empty(0, device=x.device))
        ctx.meta = (s is not None, gamma is not None, gamma.dtype if gamma is not None else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y, s, g32 = ctx.saved_tensors
        has_s, has_g, gdtype = ctx.meta
        g = g.contiguous()
        if g.dtype != y.dtype:
            g = g.to(y.dtype)
        if not has_s and not has_g:
            return g, g, None, None
        B, C = y.shape[0], y.shape[-1]
        rows = y.numel() // C
        dy = torch.empty_like(y)
        dgamma = torch.empty(C, dtype=torch.float32, device=y.device) if has_g else None
        ws = _workspace(L.load().dgtd_colsum_workspace(C), y.device) if has_g else None
        L.call("dgtd_scale_residual_bwd", L.ptr(g), L.ptr(y), L.ptr(s) if has_s else None, L.ptr(g32) if has_g else None,
               L.ptr(dy), L.ptr(dgamma), L.ptr(ws), rows, C, rows // B, L.dtype_code(y), L.stream_ptr(),
               algo=("hbm", (3 if has_g else 2) * y.element_size() * y.numel()), key=f"dgtd_scale_residual_bwd[rows={rows},C={C}]")
        return g, dy, None, (dgamma.to(gdtype) if has_g else None)


def scale_residual(x, y, s=None, gamma=None):
    if s is None and gamma is None:
        return x + y
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.scale_residual(x, y, s, gamma)
    if y.dtype != x.dtype:
        y = y.to(x.dtype)
    return _ScaleResidualFn.apply(x.contiguous(), y.contiguous(), s, gamma)


def _wgrad(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dW = dY^T X.  With M = tokens up to 131072 and N*K small, the plain GEMM has only a few hundred output tiles, each
    reducing over all M rows: latency-bound (24-100 TFLOP/s measured, tools/bench_gemm.py).  Split the token dimension
    into S batches (library batched GEMM: S times more tiles) and sum the S partial products in fp32: 5-9x faster for
    M >= 32768, 1.8x at M = 8192."""
    M = dy2.shape[0]
    S = min(32, M // 1024)
    if S < 4 or M % S or dy2.dtype == torch.float32 or not SPLITK_WGRAD:
        return dy2.t() @ x2
    part = torch.bmm(dy2.view(S, M // S, -1).transpose(1, 2), x2.view(S, M // S, -1))
    return part.sum(0)      # bf16 in, fp32 accumulation inside the reduction, bf16 out: one launch


class _LinearFn(Function):
    """y = x W^T + b with library GEMMs (hipBLASLt) and the bias gradient as ONE column-sum pass (dgtd_colsum) instead of
    a generic strided reduction per layer."""

    @staticmethod
    def forward(ctx, x, w, b):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        with torch.autocast("cuda", enabled=False):
            x2 = x.reshape(-1, x.shape[-1])
            if x2.dtype != dt:
                x2 = x2.to(dt)
            wc = w if w.dtype == dt else w.to(dt)
            # the output is allocated in its final shape and the GEMM writes into a 2-D alias of it, so what autograd sees
            # is a base tensor (a view created inside a Function may not be modified in place, e.g. by nn.ReLU(inplace=True))
            out = torch.empty(*x.shape[:-1], w.shape[0], dtype=dt, device=x.device)
            o2 = out.view(-1, w.shape[0])
            if b is not None:
                torch.addmm(b if b.dtype == dt else b.to(dt), x2, wc.t(), out=o2)
            else:
                torch.mm(x2, wc.t(), out=o2)
        ctx.save_for_backward(x2, wc)
        ctx.meta = (x.shape, x.dtype, w.dtype, b.dtype if b is not None else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, wc = ctx.saved_tensors
        xshape, xdtype, wdtype, bdtype = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        dy2 = dy2.contiguous()
        dx = (dy2 @ wc).view(xshape) if ctx.needs_input_grad[0] else None
        dw = _wgrad(dy2, x2)
        db = colsum(dy2, bdtype) if bdtype is not None else None
        return dx, (dw if dw.dtype == wdtype else dw.to(wdtype)), db


def linear(x, w, b=None):
    nat = _native.ops()
    if nat is not None and x.is_cuda and SPLITK_WGRAD:
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if dt in _CODES:
            with torch.autocast("cuda", enabled=False):
                return nat.linear(x, w, b, _CODES[dt])
    return _LinearFn.apply(x, w, b)


def _native_dt(x):
    """(native ops, dtype code) when the C++ nodes can take this call, else (None, None)."""
    nat = _native.ops()
    if nat is None or not x.is_cuda or not SPLITK_WGRAD:
        return None, None
    dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    if dt not in _CODES:
        return None, None
    return nat, _CODES[dt]


def linear_gelu(x, w, b):
    """gelu(x W^T + b) (convnext_Block pwconv1 + act, cod.py:1097-1098).  With the C++ bindings it is ONE autograd node whose
    backward fuses GELU' with the bias-gradient column sum (dgtd_gelu_bias_bwd); otherwise the composition of the separate ops."""
    nat, code = _native_dt(x)
    if nat is not None and b is not None:
        with torch.autocast("cuda", enabled=False):
            return nat.linear_gelu(x, w, b, code)
    return torch.nn.functional.gelu(linear(x, w, b))


def linear_residual(h, w, b, x, s=None, gamma=None):
    """x + s[b] * gamma[c] * (h W^T + b): Linear + layer scale + DropPath + residual (cod.py:1099-1116, :958-959).  With the C++
    bindings ONE autograd node whose backward produces dy, dgamma and the bias gradient in one pass."""
    nat, code = _native_dt(h)
    if nat is not None and b is not None:
        with torch.autocast("cuda", enabled=False):
            return nat.linear_residual(h, w, b, x, s, gamma, code)
    return scale_residual(x, linear(h, w, b), s, gamma)


def mlp_residual(v, w1, b1, w2, b2, x, s=None, gamma=None):
    """x + s[b] * gamma[c] * (gelu(v W1^T + b1) W2^T + b2): the pointwise half of a convnext_Block (cod.py:1097-1116) as ONE autograd
    node on the package's own MFMA GEMM (csrc/gemm.hip): bias + GELU and bias + layer scale + DropPath + residual are GEMM epilogues,
    and in the backward GELU' and both bias gradients are too.  Falls back to linear_gelu + linear_residual for shapes outside the kernel."""
    nat, code = _native_dt(v)
    if nat is not None and b1 is not None and b2 is not None and code != L.F32:
        M, C, H4 = v.numel() // v.shape[-1], v.shape[-1], w1.shape[0]
        if w1.dtype == v.dtype == w2.dtype and nat.gemm_ok(M, H4, C, code) and nat.gemm_ok(M, C, H4, code):
            with torch.autocast("cuda", enabled=False):
                return nat.mlp_residual(v, w1, b1, w2, b2, x, s, gamma, code)
    return linear_residual(linear_gelu(v, w1, b1), w2, b2, x, s, gamma)
