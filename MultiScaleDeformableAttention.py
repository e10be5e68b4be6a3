"""``import MultiScaleDeformableAttention as MSDA`` - the module name the reference's native extension is built under
(twig/ops/setup.py:49) and imported by (twig/ops/functions/ms_deform_attn_func.py:11, twig/ops/test.py).  The reference's module is a
CUDA-only pybind extension (setup.py:43 raises without CUDA); this one exports the same two callables (twig/ops/src/vision.cpp:13-16)
over the HIP kernels of libdgtd.so (dgtd_ms_deform_attn_fwd / _bwd, include/dgtd.h), so ms_deform_attn_func.py and test.py run
against it with no edit.  Put the repository root on sys.path (it sits next to dgtd.py)."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.abspath(__file__))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

import importlib as _importlib  # noqa: E402

import dgtd as _dgtd  # noqa: E402,F401  (registers the dgtd.* aliases of the hyphen-named package)

_impl = _importlib.import_module("dgtd.ops.ms_deform_attn")
ms_deform_attn_forward = _impl.ms_deform_attn_forward
ms_deform_attn_backward = _impl.ms_deform_attn_backward
__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
