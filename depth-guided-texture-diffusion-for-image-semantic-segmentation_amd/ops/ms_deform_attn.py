"""Multi-scale deformable attention sampling over the C ABI (csrc/ms_deform_attn.hip) — the drop-in for the reference's own native
op ``twig/ops``: same Function name, argument order and return convention as twig/ops/functions/ms_deform_attn_func.py:19-46
(forward saves its inputs, backward is once_differentiable and returns one gradient per input or None)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


def _code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.float64:
        return L.F64
    raise L.DgtdError(f"ms_deform_attn computes in float32 or float64, got {t.dtype}")


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step=64):
    """``MSDA.ms_deform_attn_forward`` of the reference's pybind module (twig/ops/src/vision.cpp:13-14, ms_deform_attn.h:20-39): raw
    tensors in, output [N, Lq, M*D] out.  ``im2col_step`` (the reference's batch chunking, ms_deform_attn_cuda.cu:44-48) has no
    counterpart - one launch covers the batch - and is accepted for signature parity."""
    value, sampling_locations, attention_weights = value.contiguous(), sampling_locations.contiguous(), attention_weights.contiguous()
    L.check_cuda(value, sampling_locations, attention_weights)
    shapes = value_spatial_shapes.to(device=value.device, dtype=torch.int64).contiguous()
    lsi = value_level_start_index.to(device=value.device, dtype=torch.int64).contiguous()
    N, S, M, D = value.shape
    _, Lq, _, Lv, P, _ = sampling_locations.shape
    if sampling_locations.dtype != value.dtype or attention_weights.dtype != value.dtype:
        raise L.DgtdError("ms_deform_attn: value, sampling_locations and attention_weights must share one dtype")
    out = torch.empty(N, Lq, M * D, dtype=value.dtype, device=value.device)
    L.call("dgtd_ms_deform_attn_fwd", L.ptr(value), L.ptr(shapes), L.ptr(lsi), L.ptr(sampling_locations), L.ptr(attention_weights),
           L.ptr(out), N, S, M, D, Lv, Lq, P, _code(value), L.stream_ptr(),
           algo=("hbm", value.element_size() * (out.numel() * (4 * Lv * P + 1))), key=f"dgtd_ms_deform_attn_fwd[N={N},Lq={Lq},M={M},D={D},L={Lv},P={P}]")
    return out


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, grad_output,
                            im2col_step=64):
    """``MSDA.ms_deform_attn_backward`` (vision.cpp:15, ms_deform_attn.h:41-60): returns the reference's
    ``[grad_value, grad_sampling_loc, grad_attn_weight]`` (ms_deform_attn_cuda.cu:149-151)."""
    value, loc, attn = value.contiguous(), sampling_locations.contiguous(), attention_weights.contiguous()
    L.check_cuda(value, loc, attn)
    shapes = value_spatial_shapes.to(device=value.device, dtype=torch.int64).contiguous()
    lsi = value_level_start_index.to(device=value.device, dtype=torch.int64).contiguous()
    N, S, M, D = value.shape
    _, Lq, _, Lv, P, _ = loc.shape
    go = grad_output.to(value.dtype).contiguous()
    gv = torch.zeros_like(value)
    gl, ga = torch.empty_like(loc), torch.empty_like(attn)
    L.call("dgtd_ms_deform_attn_bwd", L.ptr(value), L.ptr(shapes), L.ptr(lsi), L.ptr(loc), L.ptr(attn), L.ptr(go), L.ptr(gv), L.ptr(gl),
           L.ptr(ga), N, S, M, D, Lv, Lq, P, _code(value), L.stream_ptr(),
           algo=("hbm", value.element_size() * (go.numel() * (8 * Lv * P + 1))), key=f"dgtd_ms_deform_attn_bwd[N={N},Lq={Lq},M={M},D={D},L={Lv},P={P}]")
    return [gv, gl, ga]


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step=64):
        # custom_fwd(cast_inputs=torch.float32) of the reference (ms_deform_attn_func.py:21): half inputs compute in float32
        if value.dtype in (torch.float16, torch.bfloat16):
            value, sampling_locations, attention_weights = value.float(), sampling_locations.float(), attention_weights.float()
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output, ctx.im2col_step)
        return gv, None, None, gl, ga, None


def ms_deform_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step=64):
    return MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)
