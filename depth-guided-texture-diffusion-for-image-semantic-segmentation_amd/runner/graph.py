"""The training step as a captured hipGraph: forward + loss + backward + bucket gather (+ AdamW) recorded once and replayed.

Why: the eager step issues ~3300 launches from ~40 ms of host work per 43 ms of device work; a replay costs the host one
call.  What had to be true for capture to be correct (the round-1 attempt crashed in the autograd engine / replayed NaNs):

* nothing per-step may reach a launch as a HOST value: the AdamW bias corrections and learning rates are read from device memory
  (``FlatAdamW(graph_safe=True)``: ``dgtd_adamw_flat_amp`` with ``amp_state`` / ``lr_dev``), DropPath masks come from torch's
  graph-safe device generator, the loss scale lives on the device;
* the inputs are STATIC buffers the caller's batch is copied into; the FFT high-pass image (rocFFT plans and work buffers are not
  capture-safe) is computed eagerly from the static input before each replay and handed to the model (``x_hp=``);
* the eager warm-up runs on the SAME side stream the capture uses, so every leaf's AccumulateGrad node, every library handle /
  workspace (hipBLASLt plans are per thread: forward thread and autograd thread) and every MIOpen find result exists before
  capture, and gradients are ``None`` when capture starts (backward then allocates them from the graph's private pool);
* the gradient reducer does not mutate ``.grad`` from inside autograd hooks under capture (world 1: hooks only count; the gather runs
  once after backward, inside the capture);
* no torch.cat / torch.stack in the captured region (see ``_no_cat``): that, not the autograd engine, was what made the round-1
  replays produce NaNs;
* world > 1: the all-reduce stays OUTSIDE the graphs (graph A = forward/backward/gather, eager bucketed all-reduce, graph B = AdamW).
"""
from __future__ import annotations

from typing import Optional

import torch


class _no_cat:
    """torch.cat / torch.stack must not run inside the captured region: on ROCm their tensor table is staged through a pinned host
    buffer + H2D copy; a replay re-reads that host buffer after the host allocator has recycled it for some later cat, and gathers
    from stale pointers (observed: correct replays until the first eager torch.stack after a host synchronisation, then garbage
    gradients and NaNs; tools/debug_graph_nan.py).  The training path uses dgtd_multi_copy / strided copies instead
    (ops.stack, ops.cat_channels, the reducer's gather); this guard turns a regression into an error at capture time."""

    def __enter__(self):
        self._saved = (torch.cat, torch.stack)

        def refuse(*a, **k):
            raise RuntimeError("torch.cat / torch.stack inside hipGraph capture: not replay-safe on ROCm (runner/graph.py)")
        torch.cat = torch.stack = refuse
        return self

    def __exit__(self, *exc):
        torch.cat, torch.stack = self._saved
        return False


class GraphedTrainStep:
    def __init__(self, net, reducer, opt, scaler=None, warmup: int = 3):
        if not getattr(opt, "graph_safe", False):
            raise ValueError("GraphedTrainStep needs FlatAdamW(graph_safe=True): bias corrections and learning rates in device memory")
        self.net, self.reducer, self.opt, self.scaler, self.warmup = net, reducer, opt, scaler, warmup
        self.stream = torch.cuda.Stream()
        self.graph_fb: Optional[torch.cuda.CUDAGraph] = None      # forward + backward + gather (+ AdamW when world == 1)
        self.graph_opt: Optional[torch.cuda.CUDAGraph] = None     # AdamW alone (world > 1)
        self.static = None
        self.loss = None
        self.split = reducer.world > 1 or reducer._force

    # ------------------------------------------------------------------ pieces
    def _stage(self, batch) -> None:
        """Copy the batch into the static buffers and recompute the FFT high-pass image (eager, on the capture stream)."""
        s = self.static
        for k in ("input", "label", "depth"):
            v = batch[k]
            if isinstance(v, (list, tuple)):                 # mmengine's pseudo_collate: a list of per-sample tensors
                for i, t in enumerate(v):
                    s[k][i].copy_(t, non_blocking=True)
            else:
                s[k].copy_(v, non_blocking=True)
        s["x_hp"].copy_(self.net.high_pass(s["input"]))

    def _fwd_bwd(self):
        s = self.static
        self.reducer.zero_grad()
        loss = self.net(None, s["input"], s["label"], s["depth"], mode="loss", x_hp=s["x_hp"])["loss"]
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        return loss

    def _gather_all(self) -> None:
        r = self.reducer
        for b in r.buckets[r._next:]:
            r._gather(b)
        r._next = len(r.buckets)
        r._defer(False)

    def _allreduce_all(self) -> None:
        r = self.reducer
        for b in r.buckets:
            r._launch(b)
        for w in r._works:
            w.wait()
        r._works.clear()
        if r.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(r.comm_stream)

    # ------------------------------------------------------------------ capture
    def capture(self, batch) -> None:
        dev = next(self.net.parameters()).device
        stack = lambda v: (torch.stack(list(v)) if isinstance(v, (list, tuple)) else v).to(dev)
        self.static = {k: stack(batch[k]).clone() for k in ("input", "label", "depth")}
        self.static["x_hp"] = torch.empty_like(self.static["input"], dtype=torch.float32)
        overlap, self.reducer.overlap = self.reducer.overlap, False          # hooks only count; gather/launch are explicit below
        self._overlap_before = overlap
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):                                     # eager, on the capture stream
                self._stage(batch)
                self._fwd_bwd()
                self._gather_all()
                if self.split:
                    self._allreduce_all()
                self.opt.step()
            self._stage(batch)
            self.reducer.zero_grad()                                         # every leaf .grad is None when capture starts
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph_fb = torch.cuda.CUDAGraph()
        # with a process group alive, RCCL's watchdog thread keeps querying events while we capture: only this thread's (and the
        # autograd thread's, which launches into the capturing stream) calls must obey capture rules
        mode = "thread_local" if self.split else "global"
        with _no_cat(), torch.cuda.graph(self.graph_fb, stream=s, capture_error_mode=mode):
            self.loss = self._fwd_bwd()
            self._gather_all()
            if not self.split:
                self.opt.step()
        if self.split:
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, stream=s, pool=self.graph_fb.pool(), capture_error_mode=mode):
                self.opt.step()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ replay
    def __call__(self, batch) -> torch.Tensor:
        """One training step on ``batch``; returns the (static) loss tensor of this step."""
        if self.graph_fb is None:
            self.capture(batch)
        self.opt.sync_lr()
        self._stage(batch)
        self.graph_fb.replay()
        if self.split:
            self._allreduce_all()
            self.graph_opt.replay()
        return self.loss
