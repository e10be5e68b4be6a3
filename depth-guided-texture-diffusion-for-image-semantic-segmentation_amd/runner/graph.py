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
* world > 1: with RCCL the bucketed all-reduces are captured too, forked onto the reducer's side stream from the autograd hooks
  (one graph, collectives overlapped with the rest of backward); with a backend that cannot be captured (gloo) the all-reduce stays
  OUTSIDE the graphs (graph A = forward/backward/gather, eager bucketed all-reduce, graph B = AdamW).
"""
from __future__ import annotations

from typing import Optional

import torch


_L_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


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
    """``comm`` decides how the gradient all-reduce of an N > 1 job meets the captured step:

    * ``"fused"``: ONE graph with the collectives captured inside it, STAGED overlap: at the model's flush points in the backward pass
      (gradient of the diffuser embedding = Hitnet decoder + PVT + prompt decoders done; gradient of each ConvNeXt stage input) the
      weight-gradient work parked so far is flushed in whole per-stage batches, every complete bucket is gathered and its all-reduce
      forked onto the reducer's side stream, where it runs beside the rest of the backward pass; the remaining buckets follow after
      the backward pass, and AdamW walks the buckets waiting for each bucket's own all-reduce only.  Needs a backend whose
      collectives are stream-ordered and capturable (RCCL).
    * ``"hooks"``: ONE graph as well, but the autograd hooks gather a bucket when its last gradient arrives and fork its collective
      during the backward pass (the eager overlap mode, recorded).  The literal "overlap with backward" - and measured SLOWER: a gather
      inside the backward pass must flush the parked weight-gradient work early, which chops the batched launches of the deferred
      phase into per-bucket pieces and switches the shared deferrals off (profiles/r03_bench_forced_allreduce_*.json).
    * ``"split"``: graph A (forward / backward / gather) -> eager bucketed all-reduce -> graph B (AdamW); no overlap, works with any
      backend (the gloo tests).
    * ``"auto"``: fused with RCCL, split otherwise.  World 1 without DGTD_FORCE_ALLREDUCE: one graph, no collective.

    The eager warm-up steps that precede capture (handles, plans, AccumulateGrad nodes on the capture stream) run on the first batch;
    parameters, optimizer state, BatchNorm statistics, the loss scale and the RNG are snapshotted before and restored after them, so a
    captured run is step-equivalent to an eager one (``restore_after_warmup=False`` keeps them as extra training steps)."""

    def __init__(self, net, reducer, opt, scaler=None, warmup: int = 3, comm: str = "auto", restore_after_warmup: bool = True):
        if not getattr(opt, "graph_safe", False):
            raise ValueError("GraphedTrainStep needs FlatAdamW(graph_safe=True): bias corrections and learning rates in device memory")
        self.net, self.reducer, self.opt, self.scaler, self.warmup = net, reducer, opt, scaler, warmup
        self.restore = restore_after_warmup
        self.stream = torch.cuda.Stream()
        self.graph_fb: Optional[torch.cuda.CUDAGraph] = None      # forward + backward + gather (+ all-reduce + AdamW unless split)
        self.graph_opt: Optional[torch.cuda.CUDAGraph] = None     # AdamW alone (split mode)
        self.static = None
        self.loss = None
        self._pinned, self._pinned_groups = False, []
        multi = reducer.world > 1 or reducer._force
        if comm not in ("auto", "fused", "hooks", "split"):
            raise ValueError(f"comm must be auto, fused, hooks or split, got {comm!r}")
        if not multi:
            self.mode = "single"
        elif comm == "auto":
            import torch.distributed as dist
            self.mode = "fused" if dist.get_backend(reducer.group) == "nccl" else "split"
        else:
            self.mode = comm
        self.split = self.mode == "split"

    # ------------------------------------------------------------------ pieces
    def _stage(self, batch) -> None:
        """Copy the batch into the static buffers and recompute the FFT high-pass image (eager, on the capture stream)."""
        s = self.static
        for k in ("input", "label", "depth"):
            v = batch[k]
            if isinstance(v, (list, tuple)):                 # mmengine's pseudo_collate: a list of per-sample tensors
                dst = s[k]
                if (dst.dtype in _L_DTYPES and all(t.is_cuda and t.dtype == dst.dtype and t.is_contiguous() and t.numel() == dst[0].numel() for t in v)):
                    from .. import _lib as L                 # ONE launch per key instead of one copy per sample (24 serial 5-us copies at batch 8)
                    L.multi_copy(list(v), [i * dst[0].numel() for i in range(len(v))], dst.view(-1))
                    continue
                for i, t in enumerate(v):
                    dst[i].copy_(t, non_blocking=True)
            else:
                s[k].copy_(v, non_blocking=True)
        s["x_hp"].copy_(self.net.high_pass(s["input"]))

    def _fwd_bwd(self):
        s = self.static
        r = self.reducer
        staged = self.mode == "fused"
        if staged:
            r.set_staged(True)            # buckets complete at the model's flush points are all-reduced beside the rest of backward
        try:
            r.zero_grad()
            loss = self.net(None, s["input"], s["label"], s["depth"], mode="loss", x_hp=s["x_hp"])["loss"]
            (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        finally:
            if staged:
                r.set_staged(False)
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
        for b in r.buckets:
            r.wait_bucket(b["index"])
        r._works.clear()
        if r.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(r.comm_stream)

    def _one_step(self) -> torch.Tensor:
        """The step in the order the capture records it (also the eager warm-up)."""
        loss = self._fwd_bwd()
        r = self.reducer
        if self.mode == "hooks":
            r.finish()                     # hooks gathered + launched during backward; stragglers, join, fence
            self.opt.step()
        elif self.mode == "fused":
            for b in r.buckets[r._next:]:  # gather + fork the collective, bucket after bucket (the first gather flushes the deferred phase)
                r._gather(b)
                r._launch(b)
            r._next = len(r.buckets)
            r._defer(False)
            self.opt.pipelined = True
            try:
                self.opt.step()            # bucket i's AdamW waits for bucket i's all-reduce only
            finally:
                self.opt.pipelined = False
            torch.cuda.current_stream().wait_stream(r.comm_stream)     # rejoin the side stream (capture: no unjoined branch)
            r._works.clear()
        else:
            self._gather_all()
            if self.split:
                self._allreduce_all()
            self.opt.step()
        return loss

    # ------------------------------------------------------------------ state around the warm-up
    def _snapshot(self):
        r, o = self.reducer, self.opt
        snap = {"buckets": [(b["mflat"].clone(), None if b["wflat"] is None else b["wflat"].clone()) for b in r.buckets],
                "moments": [(st["exp_avg"].clone(), st["exp_avg_sq"].clone()) for st in o.state],
                "opt_state": None if o._state is None else o._state.clone(), "steps": o._steps,
                "buffers": [(bf, bf.clone()) for bf in self.net.buffers()],
                "rng": torch.cuda.get_rng_state(), "cpu_rng": torch.get_rng_state()}
        return snap

    @torch.no_grad()
    def _restore(self, snap) -> None:
        r, o = self.reducer, self.opt
        for b, (m, w) in zip(r.buckets, snap["buckets"]):
            b["mflat"].copy_(m)
            if w is not None:
                b["wflat"].copy_(w)
        for st, (m, v) in zip(o.state, snap["moments"]):
            st["exp_avg"].copy_(m)
            st["exp_avg_sq"].copy_(v)
        if snap["opt_state"] is not None:
            o._state.copy_(snap["opt_state"])
        o._steps = snap["steps"]
        for bf, v in snap["buffers"]:
            bf.copy_(v)
        torch.cuda.set_rng_state(snap["rng"])
        torch.set_rng_state(snap["cpu_rng"])

    # ------------------------------------------------------------------ capture
    def capture(self, batch) -> None:
        dev = next(self.net.parameters()).device
        stack = lambda v: (torch.stack(list(v)) if isinstance(v, (list, tuple)) else v).to(dev)
        self.static = {k: stack(batch[k]).clone() for k in ("input", "label", "depth")}
        self.static["x_hp"] = torch.empty_like(self.static["input"], dtype=torch.float32)
        r = self.reducer
        overlap = r.overlap
        self._overlap_before = overlap
        # hooks: the autograd hooks gather + fork the collectives (eager overlap mode, recorded); fused: they only mark buckets ready and
        # the flush points gather / launch (reducer.set_staged); otherwise they only count and gather / launch are explicit in _one_step
        r.overlap = overlap if self.mode in ("hooks", "fused") else False
        try:
            s = self.stream
            s.wait_stream(torch.cuda.current_stream())
            torch.cuda.synchronize()
            snap = self._snapshot() if (self.restore and self.warmup > 0) else None
            with torch.cuda.stream(s):
                for _ in range(self.warmup):                                     # eager, on the capture stream
                    self._stage(batch)
                    self._one_step()
                if snap is not None:
                    self._restore(snap)
                self._stage(batch)
                r.zero_grad()                                                    # every leaf .grad is None when capture starts
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            if self.mode != "single":
                # RCCL's watchdog thread polls the completion events of the eager collectives of the warm-up (every ~100 ms).  On ROCm 7
                # hipEventQuery refuses an event whose STREAM has meanwhile entered capture (hipErrorCapturedEvent, even though the event
                # was recorded before the capture began), and the watchdog turns that into process termination.  All warm-up work is
                # complete here (synchronize above): give the watchdog a few polling periods to retire those entries before any of the
                # process group's streams joins the capture.  (Collectives issued DURING capture are never handed to the watchdog.)
                import time
                time.sleep(0.5)
            self.graph_fb = torch.cuda.CUDAGraph()
            # with a process group alive, RCCL's watchdog thread keeps querying events while we capture: only this thread's (and the
            # autograd thread's, which launches into the capturing stream) calls must obey capture rules
            mode = "global" if self.mode == "single" else "thread_local"
            with _no_cat(), torch.cuda.graph(self.graph_fb, stream=s, capture_error_mode=mode):
                if self.split:
                    self.loss = self._fwd_bwd()
                    self._gather_all()
                else:
                    self.loss = self._one_step()
            if self.split:
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, stream=s, pool=self.graph_fb.pool(), capture_error_mode=mode):
                    self.opt.step()
            torch.cuda.synchronize()
        except BaseException:
            self.graph_fb = self.graph_opt = None
            r._works.clear()
            raise
        finally:
            r.overlap = overlap          # eager steps after (or instead of) the capture keep the reducer's own mode
        self._pin_arenas(True)

    def _pin_arenas(self, on: bool) -> None:
        """The captured graph holds raw addresses of the per-stage arenas of THIS model (csrc_torch/bindings.cpp): while pinned, a request
        that would re-allocate one of them (an eager grad-enabled step with another batch shape) raises instead of letting the next
        replay write freed memory.  Other models' arenas are not affected."""
        from ..ops import _native
        nat = _native.ops()
        if nat is None or on == self._pinned:
            return
        groups = [g for m in self.net.modules() for g in getattr(m.__dict__.get("_dgtd_arena_token"), "groups", {}).values()]
        for g in (groups if on else self._pinned_groups):
            nat.arena_pin(g, on)
        self._pinned_groups = groups if on else []
        self._pinned = on

    def release(self) -> None:
        """Drop the graphs (and un-pin the arenas they referenced)."""
        self.graph_fb = self.graph_opt = None
        self._pin_arenas(False)

    def __del__(self):
        try:
            self._pin_arenas(False)
        except Exception:
            pass

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
