"""ctypes binding of libdgtd.so (the C ABI declared in include/dgtd.h).

There is no fallback: if the library is missing or a symbol is absent, importing/using the ops
raises.  The product path never routes through PyTorch reference code or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdgtd.so")

F32, BF16, F64, F16 = 0, 1, 2, 3
_vp, _fp, _i, _i64, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); kept in lock-step with include/dgtd.h (tests/test_abi.py parses the header)
SIGNATURES = {
    "dgtd_version": (_i, []),
    "dgtd_last_error": (C.c_char_p, []),
    "dgtd_profile_enable": (_i, [_i]),
    "dgtd_profile_dump": (_i64, [C.c_char_p, _i64]),
    "dgtd_profile_empty": (_i, [_vp]),
    "dgtd_gemm_supported": (_i, [_i, _i, _i, _i]),
    "dgtd_gemm_bias": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dgtd_gemm_bias_gelu": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dgtd_gemm_bias_residual": (_i, [_vp, _vp, _vp, _vp, _fp, _fp, _vp, _vp, _i, _i, _i, _i64, _i, _vp]),
    "dgtd_gemm_gelu_bwd_workspace": (_i64, [_i, _i]),
    "dgtd_gemm_gelu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int), _i, _i, _i, _i, _vp]),
    "dgtd_transpose_batched": (_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _i, _vp]),
    "dgtd_layernorm_fwd": (_i, [_vp, _fp, _fp, _vp, _fp, _fp, _i64, _i, _f, _i, _vp]),
    "dgtd_layernorm_bwd_workspace": (_i64, [_i]),
    "dgtd_layernorm_bwd": (_i, [_vp, _vp, _fp, _fp, _fp, _vp, _fp, _fp, _vp, _i64, _i, _i, _vp]),
    "dgtd_layernorm_bwd_partial": (_i, [_vp, _vp, _fp, _fp, _fp, _vp, _vp, _vp, _i64, _i, _i, C.POINTER(C.c_int), _vp]),
    "dgtd_multi_reduce": (_i, [_vp, _i, _vp]),
    "dgtd_scale_residual_bias_bwd_partial": (_i, [_vp, _vp, _fp, _fp, _vp, _vp, _i64, _i, _i64, _i, C.POINTER(C.c_int), _vp]),
    "dgtd_gelu_bias_bwd_partial": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, C.POINTER(C.c_int), _vp]),
    "dgtd_colsum_partial": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(C.c_int), _vp]),
    "dgtd_layernorm_bwd_add": (_i, [_vp, _vp, _fp, _fp, _fp, _vp, _vp, _fp, _fp, _vp, _i64, _i, _i, _vp]),
    "dgtd_sra_attn_fwd": (_i, [_vp, _vp, _vp, _fp, _i, _i, _i, _i, _f, _i, _vp]),
    "dgtd_sra_attn_bwd_workspace": (_i64, [_i, _i, _i]),
    "dgtd_sra_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _fp, _vp, _fp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dgtd_dwconv_fwd": (_i, [_vp, _fp, _fp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_dwconv_bwd_weight_workspace": (_i64, [_i, _i, _i, _i, _i]),
    "dgtd_dwconv_bwd_weight": (_i, [_vp, _vp, _fp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_dwconv_bwd_weight_partial": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int), _vp]),
    "dgtd_dwconv_bwd_weight_batched_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "dgtd_dwconv_bwd_weight_batched": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_dwconv_pack_batched": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dgtd_dwconv_pack": (_i, [_vp, _vp, _fp, _i, _i, _i, _vp]),
    "dgtd_dwconv_unpack_grads": (_i, [_fp, _vp, _vp, _i, _i, _i, _vp]),
    "dgtd_scale_residual_fwd": (_i, [_vp, _vp, _fp, _fp, _vp, _i64, _i, _i64, _i, _vp]),
    "dgtd_colsum_workspace": (_i64, [_i]),
    "dgtd_scale_residual_bwd": (_i, [_vp, _vp, _fp, _fp, _vp, _fp, _vp, _i64, _i, _i64, _i, _vp]),
    "dgtd_colsum": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _vp]),
    "dgtd_colsum2_workspace": (_i64, [_i]),
    "dgtd_scale_residual_bias_bwd": (_i, [_vp, _vp, _fp, _fp, _vp, _fp, _vp, _i, _vp, _i64, _i, _i64, _i, _vp]),
    "dgtd_gelu_bias_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _vp]),
    "dgtd_seg_loss_workspace": (_i64, [_i, _i]),
    "dgtd_seg_loss_fwd": (_i, [_fp, _fp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "dgtd_seg_loss_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "dgtd_ssim_workspace": (_i64, []),
    "dgtd_ssim_value": (_i, [_fp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "dgtd_diffuser_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "dgtd_diffuser_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "dgtd_diffuse_tail_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "dgtd_diffuse_tail_bwd_workspace": (_i64, [_i]),
    "dgtd_diffuse_tail_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _vp, _i, _i, _vp]),
    "dgtd_adamw_flat": (_i, [_fp, _fp, _fp, _fp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "dgtd_adamw_flat_amp": (_i, [_fp, _fp, _fp, _fp, _vp, _i, _i64, _f, _f, _f, _f, _f, _f, _f, _fp, _fp, _vp]),
    "dgtd_adamw_flat_g16": (_i, [_fp, _vp, _fp, _fp, _vp, _i, _i64, _f, _f, _f, _f, _f, _f, _f, _fp, _fp, _vp]),
    "dgtd_found_inf": (_i, [_fp, _i64, _fp, _vp]),
    "dgtd_loss_scale_update": (_i, [_fp, _f, _f, _i, _vp]),
    "dgtd_im2col": (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_col2im": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_multi_copy": (_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i, _i, _vp, _i, _i, _vp]),
    "dgtd_ms_deform_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_ms_deform_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_preprocess_workspace": (_i64, [_i, _i, _i, _i]),
    "dgtd_preprocess": (_i, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_prelu_fwd": (_i, [_vp, _fp, _vp, _i64, _i, _vp]),
    "dgtd_prelu_bwd": (_i, [_vp, _vp, _fp, _vp, _fp, _i64, _i, _vp]),
    "dgtd_ca_gate_fwd": (_i, [_vp, _vp, _fp, _fp, _vp, _fp, _i, _i, _i, _i, _i, _vp]),
    "dgtd_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_bilinear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_conv3x3_supported": (_i, [_i, _i, _i, _i]),
    "dgtd_conv3x3_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_conv3x3_fwd_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_conv3x3_flip": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "dgtd_conv3x3_wgrad_workspace": (_i64, [_i, _i, _i, _i, _i, _i]),
    "dgtd_conv3x3_wgrad_batched_workspace": (_i64, [_i, _i, _i, _i, _i, _i]),
    "dgtd_conv3x3_wgrad_batched": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dgtd_ca_gate_bwd": (_i, [_vp, _vp, _fp, _fp, _fp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "dgtd_ca_gate_bwd_rows": (_i, [_vp, _vp, _fp, _fp, _fp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "dgtd_sam_supported": (_i, [_i, _i, _i, _i, _i]),
    "dgtd_sam_stats_floats": (_i64, [_i, _i, _i]),
    "dgtd_sam_scratch_floats": (_i64, [_i, _i]),
    "dgtd_sam_fwd": (_i, [_vp, _vp, _fp, _fp, _fp, _fp, _vp, _fp, _i, _i, _i, _i, _i, _vp]),
    "dgtd_sam_bwd": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "dgtd_batchnorm_supported": (_i, [_i64, _i, _i]),
    "dgtd_batchnorm_scratch": (_i64, [_i]),
    "dgtd_batchnorm_fwd": (_i, [_vp, _fp, _fp, _fp, _fp, _vp, _vp, _fp, _fp, _i64, _i, _f, _f, _i, _i, _vp]),
    "dgtd_batchnorm_bwd": (_i, [_vp, _vp, _fp, _fp, _vp, _fp, _fp, _fp, _i64, _i, _i, _vp]),
}

_lib = None


class DgtdError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for the HIP ops")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise DgtdError(f"dgtd kernels take float32, bfloat16 or float16 tensors, got {t.dtype}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def check_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise DgtdError("dgtd ops run only on the MI355X HIP device (tensor is on %s); "
                            "the CPU restatement lives in oracle/ and is test infrastructure" % t.device)
        if not t.is_contiguous():
            raise DgtdError("dgtd ops need contiguous tensors")


class Profiler:
    """HIP-event timing of every C-ABI launch on the stream it is launched on (bench.py's roofline leg).
    ``algo`` = (bound, amount): algorithmic HBM bytes or MFMA flops of that launch."""

    def __init__(self):
        self.records = []  # (key, start_event, stop_event, bound, amount)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, a, b, bound, amount in self.records:
            e = out.setdefault(key, {"calls": 0, "ms": 0.0, "amount": 0.0, "bound": bound})
            e["calls"] += 1
            e["ms"] += a.elapsed_time(b)
            e["amount"] += amount
        return out


PROFILER = None


def profile_native(on: bool) -> None:
    """Per-call device timing INSIDE libdgtd.so (dgtd_profile_enable): unlike ``PROFILER`` above it sees the calls of both host binding
    layers, i.e. the very autograd nodes the timed training step runs (C++ bindings included)."""
    load().dgtd_profile_enable(1 if on else 0)


def profile_native_summary() -> dict:
    """{key: {calls, ms, amount, bound}} of every call recorded since profile_native(True)."""
    lib = load()
    torch.cuda.synchronize()
    n = lib.dgtd_profile_dump(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.dgtd_profile_dump(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        key, bound, amount, ms = line.split("\t")
        if float(ms) < 0:
            continue
        e = out.setdefault(key, {"calls": 0, "ms": 0.0, "amount": 0.0, "bound": bound})
        e["calls"] += 1
        e["ms"] += float(ms)
        e["amount"] += float(amount)
    return out


def call(name: str, *args, algo=None, key=None):
    lib = load()
    prof = PROFILER
    if prof is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    rc = getattr(lib, name)(*args)
    if prof is not None:
        b.record()
        bound, amount = algo if algo is not None else ("hbm", 0.0)
        prof.records.append((key or name, a, b, bound, float(amount)))
    if rc != 0:
        raise DgtdError(f"{name} failed (code {rc}): {lib.dgtd_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def multi_copy(tensors, offsets, flat: torch.Tensor, to_tensors: bool = False) -> None:
    """``flat[offsets[i] : offsets[i] + tensors[i].numel()] = tensors[i]`` (or the reverse) for every i in ONE launch per 128
    tensors, converting between the tensors' dtype and ``flat``'s.  Tensors must be contiguous and share one dtype."""
    n = len(tensors)
    if n == 0:
        return
    check_cuda(flat, *tensors)
    tdt = dtype_code(tensors[0])
    P, I = C.c_void_p * n, C.c_int64 * n
    call("dgtd_multi_copy", P(*[t.data_ptr() for t in tensors]), I(*[int(o) for o in offsets]), I(*[t.numel() for t in tensors]), n, tdt,
         flat.data_ptr(), dtype_code(flat), int(to_tensors), stream_ptr())
