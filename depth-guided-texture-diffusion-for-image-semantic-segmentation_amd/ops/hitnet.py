"""Hitnet CAB decoder glue (twig/model/cod.py:415-451) over the C ABI: shared-slope PReLU and the fused
channel-attention gate + residual.  Feature maps are logical [B,C,H,W] tensors in channels_last memory (what the MIOpen
NHWC convolutions around them produce), i.e. [B,HW,C] token matrices for the kernels."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import _native


def _dense(x: torch.Tensor) -> torch.Tensor:
    if x.is_contiguous() or (x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return x
    return x.contiguous()


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous(memory_format=torch.channels_last)


def _require_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise L.DgtdError("dgtd ops run only on the MI355X HIP device (tensor is on %s)" % t.device)


class _PReLUFn(Function):
    @staticmethod
    def forward(ctx, x, a):
        _require_cuda(x)
        x = _dense(x)
        a32 = a.detach().float().contiguous()
        y = torch.empty_like(x)
        L.call("dgtd_prelu_fwd", L.ptr(x), L.ptr(a32), L.ptr(y), x.numel(), L.dtype_code(x), L.stream_ptr(),
               algo=("hbm", 2 * x.element_size() * x.numel()), key=f"dgtd_prelu_fwd[n={x.numel()}]")
        ctx.save_for_backward(x, a32)
        ctx.adtype = a.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, a32 = ctx.saved_tensors
        if g.dtype != x.dtype or g.stride() != x.stride():
            g = torch.empty_like(x).copy_(g)
        dx = torch.empty_like(x)
        da = torch.zeros(1, dtype=torch.float32, device=x.device)
        L.call("dgtd_prelu_bwd", L.ptr(x), L.ptr(g), L.ptr(a32), L.ptr(dx), L.ptr(da), x.numel(), L.dtype_code(x),
               L.stream_ptr(), algo=("hbm", 3 * x.element_size() * x.numel()), key=f"dgtd_prelu_bwd[n={x.numel()}]")
        return dx, da.to(ctx.adtype)


def prelu(x: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """nn.PReLU() with a single shared slope (cod.py:686; applied inside every CAB, cod.py:444-446)."""
    if a.numel() != 1:
        raise L.DgtdError("dgtd prelu implements the single-slope nn.PReLU() the reference constructs (cod.py:686)")
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.prelu(x, a)
    return _PReLUFn.apply(x, a)


class _CAGateFn(Function):
    @staticmethod
    def forward(ctx, res, x, w1, w2):
        _require_cuda(res, x)
        res, x = _nhwc(res), _nhwc(x if x.dtype == res.dtype else x.to(res.dtype))
        B, C, H, W = res.shape
        R = w1.shape[0]
        w1f = w1.detach().reshape(R, C).float().contiguous()
        w2f = w2.detach().reshape(C, R).float().contiguous()
        stats = torch.empty(2 * B * C + B * R + 64 * B * C, dtype=torch.float32, device=res.device)
        out = torch.empty_like(res)
        L.call("dgtd_ca_gate_fwd", L.ptr(res), L.ptr(x), L.ptr(w1f), L.ptr(w2f), L.ptr(out), L.ptr(stats), B, H * W, C, R,
               L.dtype_code(res), L.stream_ptr(), algo=("hbm", 4 * res.element_size() * res.numel()),
               key=f"dgtd_ca_gate_fwd[B={B},HW={H * W},C={C}]")
        ctx.save_for_backward(res, w1f, w2f, stats)
        ctx.meta = (w1.shape, w1.dtype, w2.shape, w2.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        res, w1f, w2f, stats = ctx.saved_tensors
        w1shape, w1dtype, w2shape, w2dtype = ctx.meta
        B, C, H, W = res.shape
        R = w1f.shape[0]
        g = _nhwc(g if g.dtype == res.dtype else g.to(res.dtype))
        dres = torch.empty_like(res)
        small = torch.empty(2 * R * C + B * C + 64 * B * C + B * 2 * R * C, dtype=torch.float32, device=res.device)
        dw1, dw2, scratch = small[:R * C], small[R * C:2 * R * C], small[2 * R * C:]
        L.call("dgtd_ca_gate_bwd", L.ptr(g), L.ptr(res), L.ptr(w1f), L.ptr(w2f), L.ptr(stats), L.ptr(dres), dw1.data_ptr(),
               dw2.data_ptr(), scratch.data_ptr(), B, H * W, C, R, L.dtype_code(res), L.stream_ptr(),
               algo=("hbm", 4 * res.element_size() * res.numel()), key=f"dgtd_ca_gate_bwd[B={B},HW={H * W},C={C}]")
        return dres, g, dw1.view(w1shape).to(w1dtype), dw2.view(w2shape).to(w2dtype)


def ca_gate(res: torch.Tensor, x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """res * sigmoid(conv1x1_w2(relu(conv1x1_w1(avgpool(res))))) + x — CALayer (cod.py:428-431) and the CAB residual (cod.py:451).
    w1 [C/r, C, 1, 1], w2 [C, C/r, 1, 1] are the bias-free conv_du weights (cod.py:421-425)."""
    nat = _native.ops()
    if nat is not None and res.is_cuda:
        return nat.ca_gate(res, x, w1, w2)
    return _CAGateFn.apply(res, x, w1, w2)


class _BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        _require_cuda(x)
        x = _nhwc(x)
        B, C, Hi, Wi = x.shape
        y = torch.empty(B, C, Ho, Wo, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.call("dgtd_bilinear_fwd", L.ptr(x), L.ptr(y), B, Hi, Wi, Ho, Wo, C, int(align), L.dtype_code(x), L.stream_ptr(),
               algo=("hbm", x.element_size() * (x.numel() + y.numel())), key=f"dgtd_bilinear_fwd[{Hi}x{Wi}->{Ho}x{Wo},C={C}]")
        ctx.meta = (B, C, Hi, Wi, Ho, Wo, align, x.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, C, Hi, Wi, Ho, Wo, align, dtype = ctx.meta
        g = _nhwc(g if g.dtype == dtype else g.to(dtype))
        dx = torch.empty(B, C, Hi, Wi, dtype=dtype, device=g.device, memory_format=torch.channels_last)
        L.call("dgtd_bilinear_bwd", L.ptr(g), L.ptr(dx), B, Hi, Wi, Ho, Wo, C, int(align), L.dtype_code(g), L.stream_ptr(),
               algo=("hbm", g.element_size() * (g.numel() + dx.numel())), key=f"dgtd_bilinear_bwd[{Hi}x{Wi}<-{Ho}x{Wo},C={C}]")
        return dx, None, None, None


def bilinear_resize(x: torch.Tensor, out_h: int, out_w: int, align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size=(out_h, out_w), mode="bilinear", align_corners=...) for a logical [B,C,H,W] map kept channels_last."""
    nat = _native.ops()
    if nat is not None and x.is_cuda:
        return nat.bilinear_resize(x, int(out_h), int(out_w), bool(align_corners))
    return _BilinearFn.apply(x, int(out_h), int(out_w), bool(align_corners))


class _StackFn(Function):
    """torch.stack(tensors, 0) for same-shape contiguous tensors as ONE dgtd_multi_copy launch (hipGraph-safe, see csrc/multicopy.hip);
    the backward hands each input its slice of the gradient (views, no kernel)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [x.contiguous() for x in xs]
        out = torch.empty((len(xs),) + tuple(xs[0].shape), dtype=xs[0].dtype, device=xs[0].device)
        n = xs[0].numel()
        L.multi_copy(xs, [i * n for i in range(len(xs))], out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return tuple(g.unbind(0))


def stack(tensors) -> torch.Tensor:
    tensors = list(tensors)
    if not tensors[0].is_cuda or any(t.dtype != tensors[0].dtype for t in tensors):
        return torch.stack(tensors)
    return _StackFn.apply(*tensors)


class _CatChannelsFn(Function):
    """torch.cat(maps, dim=1) for channels_last maps [B,C_i,H,W] (the skip concatenations of the Hitnet decoder, cod.py:777-792, and
    the four taps of ShapePropEncoder, cod.py:1176): one strided copy per input into a channels_last output, gradient = channel
    slices (views).  No torch.cat: see csrc/multicopy.hip on why it cannot be captured in a hipGraph on ROCm."""

    @staticmethod
    def forward(ctx, *xs):
        B, _, H, W = xs[0].shape
        ctx.splits = [x.shape[1] for x in xs]
        out = torch.empty((B, sum(ctx.splits), H, W), dtype=xs[0].dtype, device=xs[0].device, memory_format=torch.channels_last)
        off = 0
        for x in xs:
            out.narrow(1, off, x.shape[1]).copy_(x)
            off += x.shape[1]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.splits:
            outs.append(g.narrow(1, off, c))
            off += c
        return tuple(outs)


def cat_channels(maps) -> torch.Tensor:
    maps = list(maps)
    if not maps[0].is_cuda:
        return torch.cat(maps, dim=1)
    dt = maps[0].dtype
    return _CatChannelsFn.apply(*[m if m.dtype == dt else m.to(dt) for m in maps])


class _UnstackFn(Function):
    """x [Z, ...] -> Z views x[z] (torch.unbind), with ONE gather kernel in the backward.  Slicing / indexing the stacked trunk
    output with plain autograd costs a full-size zero-filled tensor plus an add per consumer (SelectBackward / SliceBackward:
    ~40 launches and >1 GB of traffic per step for the 16 prompt decoders), and unbind's own backward is at::stack (not
    hipGraph-safe on ROCm, csrc/multicopy.hip)."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        return tuple(x.unbind(0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        shape, dtype, device = ctx.meta
        out = torch.empty(shape, dtype=dtype, device=device)
        n = out[0].numel()
        live = [(i, (g if g.dtype == dtype else g.to(dtype)).contiguous()) for i, g in enumerate(gs) if g is not None]
        for i, g in enumerate(gs):
            if g is None:
                out[i].zero_()
        L.multi_copy([g for _, g in live], [i * n for i, _ in live], out)
        return out


def unstack(x: torch.Tensor):
    return _UnstackFn.apply(x) if x.is_cuda else tuple(x.unbind(0))


def cab(x: torch.Tensor, w0: torch.Tensor, w1: torch.Tensor, a: torch.Tensor, cw1: torch.Tensor, cw2: torch.Tensor):
    """The whole CAB (cod.py:436-451) - conv3x3 -> PReLU -> conv3x3 -> channel-attention gate -> + x - as ONE autograd node of the C++
    binding layer (csrc_torch/bindings.cpp CabFn): the PReLU and its backward ride in the convolutions' epilogues
    (dgtd_conv3x3_fwd_ex), the skip gradient is added inside the first convolution's input-gradient launch.  None when the node is not
    available (Python bindings, profiler attached): the caller composes the separate ops."""
    nat = _native.ops()
    if nat is None or not x.is_cuda or not hasattr(nat, "cab"):
        return None
    return nat.cab(x, w0, w1, a, cw1, cw2)


class _SamFn(Function):
    @staticmethod
    def forward(ctx, xh, xl, w1, w2, v1, v2):
        _require_cuda(xh, xl)
        xh, xl = _nhwc(xh), _nhwc(xl if xl.dtype == xh.dtype else xl.to(xh.dtype))
        B, C, H, W = xh.shape
        R = w1.shape[0]
        ws = [w1.detach().reshape(R, C).float().contiguous(), w2.detach().reshape(C, R).float().contiguous(),
              v1.detach().reshape(R, C).float().contiguous(), v2.detach().reshape(R).float().contiguous()]
        stats = torch.empty(L.load().dgtd_sam_stats_floats(B, C, R), dtype=torch.float32, device=xh.device)
        out = torch.empty_like(xh)
        L.call("dgtd_sam_fwd", L.ptr(xh), L.ptr(xl), *(L.ptr(w) for w in ws), L.ptr(out), L.ptr(stats), B, H * W, C, R, L.dtype_code(xh),
               L.stream_ptr(), algo=("hbm", 5 * xh.element_size() * xh.numel()), key=f"dgtd_sam_fwd[B={B},HW={H * W},C={C}]")
        ctx.save_for_backward(xh, xl, *ws, stats)
        ctx.meta = [(w.shape, w.dtype) for w in (w1, w2, v1, v2)]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xh, xl, w1f, w2f, v1f, v2f, stats = ctx.saved_tensors
        B, C, H, W = xh.shape
        R = w1f.shape[0]
        g = _nhwc(g if g.dtype == xh.dtype else g.to(xh.dtype))
        dxh, dxl = torch.empty_like(xh), torch.empty_like(xh)
        dw = torch.empty(3 * R * C + R, dtype=torch.float32, device=xh.device)
        scratch = torch.empty(L.load().dgtd_sam_scratch_floats(B, C), dtype=torch.float32, device=xh.device)
        L.call("dgtd_sam_bwd", L.ptr(g), L.ptr(xh), L.ptr(xl), L.ptr(w1f), L.ptr(w2f), L.ptr(v1f), L.ptr(v2f), L.ptr(stats), L.ptr(dxh), L.ptr(dxl),
               L.ptr(dw), L.ptr(scratch), B, H * W, C, R, L.dtype_code(xh), L.stream_ptr(),
               algo=("hbm", 7 * xh.element_size() * xh.numel()), key=f"dgtd_sam_bwd[B={B},HW={H * W},C={C}]")
        parts = (dw[:R * C], dw[R * C:2 * R * C], dw[2 * R * C:3 * R * C], dw[3 * R * C:])
        return (dxh, dxl) + tuple(p.view(shape).to(dtype) for p, (shape, dtype) in zip(parts, ctx.meta))


def sam_supported(x: torch.Tensor, R: int) -> bool:
    return bool(x.is_cuda and x.ndim == 4 and
                L.load().dgtd_sam_supported(x.shape[0], x.shape[2] * x.shape[3], x.shape[1], R, L.dtype_code(x)))


def sam(xh: torch.Tensor, xl: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
    """SAM (cod.py:454-506): x_h * fc(mean x_h) * fc_wight(mean x_h) + x_l * fc(mean x_l) * fc_wight(mean x_l) in two launches
    (backward: two).  w1 [R,C], w2 [C,R] = SAM.fc's bias-free Linear weights (cod.py:459-464); v1 [R,C], v2 [1,R] = SAM.fc_wight's
    (cod.py:465-470)."""
    nat = _native.ops()
    if nat is not None and xh.is_cuda and hasattr(nat, "sam"):
        return nat.sam(xh, xl, w1, w2, v1, v2)
    return _SamFn.apply(xh, xl, w1, w2, v1, v2)


class _BatchNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, training, momentum, eps):
        _require_cuda(x)
        x = _nhwc(x)
        C = x.shape[1]
        N = x.numel() // C
        y = torch.empty_like(x)
        save = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        scratch = torch.empty(L.load().dgtd_batchnorm_scratch(C), dtype=torch.float32, device=x.device)
        L.call("dgtd_batchnorm_fwd", L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), L.ptr(num_batches), L.ptr(y),
               L.ptr(save), L.ptr(scratch), N, C, float(eps), float(momentum), int(training), L.dtype_code(x), L.stream_ptr(),
               algo=("hbm", (3 if training else 2) * x.element_size() * x.numel()),
               key=f"dgtd_batchnorm_fwd[{'train' if training else 'eval'},N={N},C={C}]")
        ctx.save_for_backward(x, gamma, save)
        ctx.training = bool(training)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, gamma, save = ctx.saved_tensors
        if not ctx.training:
            raise L.DgtdError("dgtd batch_norm: the backward exists for training statistics only")
        C = x.shape[1]
        N = x.numel() // C
        g = _nhwc(g if g.dtype == x.dtype else g.to(x.dtype))
        dx = torch.empty_like(x)
        d = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        scratch = torch.empty(L.load().dgtd_batchnorm_scratch(C), dtype=torch.float32, device=x.device)
        L.call("dgtd_batchnorm_bwd", L.ptr(g), L.ptr(x), L.ptr(gamma), L.ptr(save), L.ptr(dx), d[:C].data_ptr(), d[C:].data_ptr(), L.ptr(scratch),
               N, C, L.dtype_code(x), L.stream_ptr(), algo=("hbm", 5 * x.element_size() * x.numel()), key=f"dgtd_batchnorm_bwd[N={N},C={C}]")
        return dx, d[:C], d[C:], None, None, None, None, None, None


def batch_norm_supported(x: torch.Tensor, bn: torch.nn.BatchNorm2d) -> bool:
    """True when ``bn(x)`` can run on the fused kernels: affine fp32 parameters, tracked statistics with a fixed momentum, training
    mode or no gradient wanted, and a channel count the kernels tile (a power of two in [8, 128])."""
    if not (x.is_cuda and x.ndim == 4 and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32):
        return False
    if not bn.training and torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
        return False
    return bool(L.load().dgtd_batchnorm_supported(x.numel() // x.shape[1], x.shape[1], L.dtype_code(x)))


def batch_norm(x: torch.Tensor, bn: torch.nn.BatchNorm2d) -> torch.Tensor:
    """``bn(x)`` for the nn.BatchNorm2d of a BasicConv2d (cod.py:359, :366) on a channels_last map: two launches forward (the running
    statistics and num_batches_tracked move inside the second), two backward.  Same state_dict entries, same update rule as torch's."""
    args = (x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bool(bn.training), float(bn.momentum), float(bn.eps))
    nat = _native.ops()
    if nat is not None and hasattr(nat, "batch_norm"):
        return nat.batch_norm(*args)
    return _BatchNormFn.apply(*args)
