"""dgtd — MI355X-native hot path of *Depth-guided Texture Diffusion for Image Semantic Segmentation*.

Scope: the forward/backward of the reference's `cod` model (twig/model/cod.py) as
drop-in nn.Modules whose hot ops are hand-written gfx950 HIP kernels behind the C ABI of
include/dgtd.h (libdgtd.so, bound with ctypes in `_lib`), plus data-parallel gradient reduction
over RCCL (`dist`).  PyTorch-ROCm supplies device memory, streams, autograd and library GEMM/conv.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401
from . import nn  # noqa: F401
from . import dist  # noqa: F401
from . import runner  # noqa: F401

__all__ = ["_lib", "ops", "nn", "dist", "runner"]
