"""Optional C++ autograd bindings (libdgtd_torch.so, csrc_torch/bindings.cpp) over the same C ABI.  They only remove Python
overhead from the host-bound training step; kernels, numerics and error behaviour are those of libdgtd.so either way.  The Python
autograd.Functions remain the reference binding and are used whenever the HIP-event profiler is active (bench.py roofline leg)
or when DGTD_TORCH_BINDINGS=0."""
from __future__ import annotations

import os

import torch

from .. import _lib as L

TORCH_LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdgtd_torch.so")
ENABLED = os.environ.get("DGTD_TORCH_BINDINGS", "1") != "0"
_ops = None
_tried = False


def ops():
    """torch.ops.dgtd when the binding library is built and enabled and no profiler is attached, else None."""
    global _ops, _tried
    if not ENABLED or L.PROFILER is not None:
        return None
    if not _tried:
        _tried = True
        if os.path.exists(TORCH_LIB_PATH):
            L.load()                                   # libdgtd.so first (the bindings link against it)
            torch.ops.load_library(TORCH_LIB_PATH)
            _ops = torch.ops.dgtd
    return _ops
