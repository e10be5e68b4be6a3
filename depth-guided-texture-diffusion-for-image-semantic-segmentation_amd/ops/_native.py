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


# ------------------------------------------------------------------------------------------------ runs of identical blocks
_NEXT_GROUP = [0]
BATCH_WGRAD = os.environ.get("DGTD_BATCH_WGRAD", "1") != "0"


class _Released:
    """Frees the arenas of an owner module's groups when the module is garbage-collected."""

    def __init__(self):
        self.groups = {}       # stage -> arena group id

    def __del__(self):
        nat = _ops
        if nat is not None:
            for g in self.groups.values():
                try:
                    nat.arena_release(g)
                except Exception:
                    pass


class block_run:
    """``with block_run(owner, stage, n_blocks, x) as run: for j, blk in ...: run.at(j); x = blk(x)``

    Tells the C++ nodes that the calls inside belong to block j of a run of n identical blocks (a ConvNeXt stage): under a gradient
    reducer (deferral on, 16-bit compute) the producers of the Linear layers' inputs and output gradients then write into per-stage
    arenas, and the n weight-gradient GEMMs of each Linear of the run become ONE strided-batched GEMM at the end of the backward pass
    (bindings.cpp "Deferred WEIGHT GRADIENTS of the Linear layers").  A no-op otherwise (CPU, fp32, Python bindings, eval)."""

    def __init__(self, owner, stage: int, n: int, x: torch.Tensor):
        self.nat = ops() if (BATCH_WGRAD and x.is_cuda and n > 1 and torch.is_grad_enabled()) else None
        self.n = n
        if self.nat is not None:
            tok = owner.__dict__.get("_dgtd_arena_token")          # lives and dies with the owner module
            if tok is None:
                tok = _Released()
                object.__setattr__(owner, "_dgtd_arena_token", tok)
            if stage not in tok.groups:
                tok.groups[stage] = _NEXT_GROUP[0]
                _NEXT_GROUP[0] += 1
            self.group = tok.groups[stage]

    def __enter__(self):
        return self

    def at(self, j: int) -> None:
        if self.nat is not None:
            self.nat.arena_hint(self.group, j, self.n)

    def roles(self, out: int = -1, grad_a: int = -1, grad_b: int = -1) -> None:
        """The NEXT arena-capable node writes its forward output as the input X[out] of the block's out-th deferred Linear, and
        the gradients its backward produces as DY[grad_a] (, DY[grad_b]): the output gradients of those Linears."""
        if self.nat is not None:
            self.nat.arena_roles(out, grad_a, grad_b)

    def __exit__(self, *exc):
        if self.nat is not None:
            self.nat.arena_hint(-1, 0, 0)
        return False


class _NoRun:
    def at(self, j):
        pass

    def roles(self, out=-1, grad_a=-1, grad_b=-1):
        pass


NO_RUN = _NoRun()


# ------------------------------------------------------------------------------------------------ flush points
# Places in the BACKWARD pass where every gradient produced so far is complete and no shared deferral is half-way: when the gradient of
# the texture-diffuser embedding arrives (Hitnet decoder, all PVT blocks and the prompt decoders are done) and when the gradient of a
# ConvNeXt stage's input arrives (that stage is done).  A gradient reducer in staged mode (dist.GradReducer.staged) registers its
# flush_point() here: it flushes the deferred weight-gradient work parked so far (whole per-stage batches), gathers every complete
# bucket and forks its all-reduce, which then runs beside the rest of the backward pass.
FLUSH_POINT = [None]


def flush_point_hook(grad):
    fn = FLUSH_POINT[0]
    if fn is not None:
        fn()
    return None


def mark_flush_point(t: torch.Tensor) -> None:
    """Call in the forward pass on a tensor whose gradient marks a flush point (no-op unless a staged reducer is listening)."""
    if FLUSH_POINT[0] is not None and t.requires_grad:
        t.register_hook(flush_point_hook)
