"""Bucketed gradient handling for the single-node data-parallel training step.

The reference gets this from mmengine's MMDistributedDataParallel with find_unused_parameters=True
(config/sod.yml:11) plus AmpOptimWrapper (config/sod.yml:57): a per-step graph walk, a per-step BN-buffer
broadcast, one cast kernel per weight per forward and one accumulate kernel per parameter per backward.
Here, per bucket (reverse registration order ≈ gradient readiness: Hitnet heads -> PVT stage 4..1 -> prompt
decoders -> ConvNeXt 3..0 -> diffuser 1x1s):
  * fp32 MASTER parameters stay the modules' registered parameters (state_dict contract, optimizer state);
  * Linear/Conv2d weights and biases get a low-precision WORKING COPY (leaf tensors inside one flat buffer),
    refreshed with ONE multi-tensor cast per bucket after the optimizer step; the modules compute with it;
  * autograd hands every leaf its gradient without an accumulate kernel (``.grad`` is None before backward);
    when the last leaf of a bucket is ready, ONE multi-tensor copy casts the bucket's gradients into a flat
    fp32 buffer, and (N > 1) ONE all-reduce is launched on a side HIP stream fenced by events, so RCCL traffic
    over xGMI overlaps the rest of the backward;  master ``.grad``s are views of that flat buffer;
  * the parameters that never receive a gradient (5 tensors) are excluded statically;
  * BatchNorm statistics stay per-rank (plain BatchNorm2d, cod.py:362) and are not broadcast.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

# constructed but never called in the reference forward (cod.py:703-704, :1251): no gradient, ever
STATIC_UNUSED = ("hitnet.backbone.prompt_encoder.adaptor.", "hitnet.ca.", "hitnet.sa.")
DEFER = os.environ.get("DGTD_DEFER_REDUCTIONS", "1") != "0"   # A/B switch: batched second-stage column reductions
ALIGN = 8   # elements: every tensor of a bucket starts on a 16-byte (2-byte dtypes) / 32-byte (fp32) boundary


def init_process_group(backend: Optional[str] = None) -> tuple:
    """One process per GPU; rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("DGTD_DIST_BACKEND") == "gloo" and torch.cuda.is_available() and torch.cuda.device_count() == 1:
        local = 0      # several gloo ranks sharing the one GPU of a test box
    # DGTD_FORCE_ALLREDUCE=1: rehearse the RCCL path (process group, side-stream all-reduce) with a single rank on a one-GPU box
    if (world > 1 or os.environ.get("DGTD_FORCE_ALLREDUCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # DGTD_DIST_BACKEND=gloo: several ranks on ONE GPU (tests of the multi-rank control flow on a one-GPU box; RCCL needs a GPU per rank)
        backend = backend or os.environ.get("DGTD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """One-time replica sync (parameters AND buffers) from rank ``src``; coalesced into few large messages."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradReducer:
    """Flat gradient buckets (+ optional low-precision working weights) for ``module``."""

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 64 << 20, group=None,
                 exclude_prefixes: Sequence[str] = STATIC_UNUSED, overlap: bool = True,
                 working_dtype: Optional[torch.dtype] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.working_dtype = working_dtype if working_dtype not in (None, torch.float32) else None
        # parameters that the modules consume through wb(): weights / biases of the package's Linear / Conv2d
        castable = {}
        if self.working_dtype is not None:
            from ..nn.modules import Conv2d, Linear
            for m in module.modules():
                if isinstance(m, (Linear, Conv2d)) and not getattr(m, "keep_master", False):
                    castable[id(m.weight)] = (m, "_w")
                    if m.bias is not None:
                        castable[id(m.bias)] = (m, "_b")
        seen, params = set(), []
        for n, p in module.named_parameters():  # named_parameters() de-duplicates the shared PReLU
            if not p.requires_grad or id(p) in seen or any(n.startswith(e) for e in exclude_prefixes):
                continue
            seen.add(id(p))
            params.append((n, p))
        params.reverse()  # readiness order ≈ reverse registration order
        # ... refined by the model where registration order and backward order disagree: dgtd.nn.cod registers the texture diffuser
        # (ConvNeXt trunk) BEFORE the prompt decoders but AFTER the PVT stages, while in the backward pass the PVT stages and the prompt
        # decoders finish first and the ConvNeXt trunk last.  Buckets go out strictly in order, so without this every PVT bucket would
        # queue behind the ConvNeXt buckets.  (Stable sort: reverse registration order inside a rank; identical on every rank.)
        rank_of = getattr(module, "grad_readiness_rank", None)
        if callable(rank_of):
            params.sort(key=lambda np_: rank_of(np_[0]))
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for n, p in params:
            cur.append((n, p))
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._seal(cur, castable)
                cur, cur_bytes = [], 0
        if cur:
            self._seal(cur, castable)
        self._force = os.environ.get("DGTD_FORCE_ALLREDUCE") == "1" and dist.is_initialized()
        self.overlap = overlap and (self.world > 1 or self._force)
        # N > 1: the working-copy segment of a bucket travels in the WORKING dtype (its leaf gradients are 16-bit tensors already, so
        # the gather into a 16-bit buffer is lossless; 228.8 MB instead of 457.7 MB over xGMI at config 2) and is widened into the
        # fp32 bucket on the comm stream right after its all-reduce.  DGTD_COMM_FP32=1 keeps the fp32 payload.
        self.comm16 = (self.working_dtype is not None and (self.world > 1 or self._force)
                       and os.environ.get("DGTD_COMM_FP32", "0") != "1")
        if self.comm16:
            for b in self.buckets:
                if b["n_work"]:
                    b["g16"] = torch.zeros(b["n_work"], dtype=self.working_dtype, device=b["flat"].device)
        self._cuda = bool(self.buckets) and self.buckets[0]["flat"].is_cuda
        self.comm_stream = torch.cuda.Stream() if (self._cuda and self.overlap) else None
        self._works, self._next = [], 0
        self.staged = False
        for b in self.buckets:
            for leaf in b["leaves"]:
                leaf.register_post_accumulate_grad_hook(self._make_hook(b))
        self._install_load_hooks()
        self.refresh_working()

    # ------------------------------------------------------------------ construction
    def _seal(self, items, castable) -> None:
        """Flat layout of a bucket: [ castable parameters | the rest ].  The castable masters are re-homed into one
        flat fp32 buffer (``p.data`` becomes a view), so master -> working copy is ONE cast kernel.
        Every tensor starts at a multiple of ALIGN elements (16 B in the 2-byte working copy, 32 B in fp32): a bias of one
        element (out_CFM / out_SAM) would otherwise leave every later weight, gradient and AdamW-state slice at an odd
        byte phase, where 16-byte vector accesses are split.  The padding is zero in every buffer, so AdamW on it is a no-op."""
        dev = items[0][1].device
        work = [(n, p) for n, p in items if id(p) in castable]
        rest = [(n, p) for n, p in items if id(p) not in castable]
        pad_to = lambda n: -(-n // ALIGN) * ALIGN
        n_work, n_rest = sum(pad_to(p.numel()) for _, p in work), sum(pad_to(p.numel()) for _, p in rest)
        flat = torch.zeros(n_work + n_rest, dtype=torch.float32, device=dev)          # fp32 gradients
        mflat = torch.zeros(n_work + n_rest, dtype=torch.float32, device=dev)            # fp32 masters of the whole bucket (flat: one AdamW launch per run)
        wflat = torch.zeros(n_work, dtype=self.working_dtype, device=dev) if n_work else None
        masters, leaves, gviews, nhwc, offsets, modules = [], [], [], [], [], []
        off = 0
        with torch.no_grad():
            for _, p in work + rest:
                n = p.numel()
                # Dense KxK convolution kernels live in O,H,W,I storage order (channels_last strides, logical shape unchanged):
                # the NHWC convolutions then take the working copy as it is instead of re-laying it out on every call, and the
                # weight gradient they return lands in the bucket without a layout copy.  k == stride convs are consumed as
                # GEMM matrices through weight.flatten(1) and keep the O,I,H,W order.
                cl = False
                if id(p) in castable and p.ndim == 4 and p.shape[1] > 1 and p.shape[2] * p.shape[3] > 1:
                    m = castable[id(p)][0]
                    cl = tuple(m.stride) != tuple(m.kernel_size)
                nhwc.append(cl)
                offsets.append(off)
                view = (lambda t: t.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)) if cl \
                    else (lambda t: t.view_as(p))
                gv = view(flat[off:off + n])
                p.grad = gv                      # master .grad = view of the flat fp32 bucket (what the optimizer reads)
                masters.append(p)
                gviews.append(gv)
                if id(p) in castable:
                    m, attr = castable[id(p)]
                    mv = view(mflat[off:off + n])
                    mv.copy_(p.data)
                    p.data = mv
                    leaf = view(wflat[off:off + n])
                    leaf.requires_grad_(True)    # a leaf: its base buffer does not require grad
                    object.__setattr__(m, attr, leaf)
                    p.requires_grad_(False)      # the master no longer takes part in autograd
                    leaves.append(leaf)
                    modules.append((m, attr))
                else:
                    mv = mflat[off:off + n].view_as(p)   # re-homed as well: the flat optimizer updates the bucket in place
                    mv.copy_(p.data)
                    p.data = mv
                    leaves.append(p)
                    modules.append(None)
                off += pad_to(n)
        self.buckets.append({"flat": flat, "mflat": mflat, "wflat": wflat, "n_work": n_work, "k_work": len(work),
                             "masters": masters, "leaves": leaves, "gviews": gviews, "nhwc": nhwc, "offsets": offsets,
                             "names": [n_ for n_, _ in work + rest], "sizes": [p_.numel() for _, p_ in work + rest],
                             "padded": [pad_to(p_.numel()) for _, p_ in work + rest], "shapes": [tuple(p_.shape) for _, p_ in work + rest],
                             "modules": modules, "missing": (), "pads": {}, "g16": None,
                             "pending": len(items), "n": len(items), "done": False, "ready": False, "index": len(self.buckets)})

    # ------------------------------------------------------------------ per step
    @torch.no_grad()
    def refresh_working(self) -> None:
        """master fp32 -> working copy: ONE cast kernel per bucket (call after optimizer.step())."""
        for b in self.buckets:
            if b["wflat"] is not None:
                b["wflat"].copy_(b["mflat"][:b["n_work"]])

    def _install_load_hooks(self) -> None:
        """``load_state_dict`` copies into the fp32 masters only; the modules compute with the working copies.  Every module
        that owns a working copy refreshes it right after its own parameters were loaded (whichever (sub)module the caller
        loaded through: runner.load_pretrained loads ``hitnet.backbone`` and ``encoder2`` directly)."""
        def hook(module, _incompatible):
            with torch.no_grad():
                for attr, src in (("_w", module.weight), ("_b", module.bias)):
                    leaf = getattr(module, attr, None)
                    if leaf is not None and src is not None:
                        leaf.copy_(src)
        seen = set()
        for b in self.buckets:
            for ma in b["modules"]:
                if ma is not None and id(ma[0]) not in seen:
                    seen.add(id(ma[0]))
                    ma[0].register_load_state_dict_post_hook(hook)

    def zero_grad(self) -> None:
        """Replaces optimizer.zero_grad(): leaves get .grad = None so autograd hands gradients over without an
        accumulate kernel; nothing is memset (the flat buckets are fully overwritten by _gather)."""
        self._next = 0
        for b in self.buckets:
            b["pending"], b["done"], b["ready"] = b["n"], False, False
            for leaf in b["leaves"]:
                leaf.grad = None
        self._defer(True)

    def _defer(self, on: bool) -> None:
        """Bracket backward(): while on, the C++ backward nodes park the second stage of their column reductions (LayerNorm
        dgamma/dbeta, Linear bias gradients) and ONE dgtd_multi_reduce per 56 of them runs at the first gather (csrc_torch/bindings.cpp)."""
        if self._cuda and DEFER:
            from ..ops import _native
            nat = _native.ops()
            if nat is not None:
                if on:      # one gradient tensor for all calls of a shared weight needs a single flush, after the whole backward pass
                    nat.set_shared_deferral(not self.overlap or self.staged)   # staged: flushes happen only at the model's flush points
                nat.set_deferred(on)

    def _flush_deferred(self) -> None:
        if self._cuda and DEFER:
            from ..ops import _native
            nat = _native.ops()
            if nat is not None:
                nat.flush_deferred()

    def _make_hook(self, bucket):
        def hook(_leaf):
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                bucket["ready"] = True
                if self.overlap and not self.staged:
                    self._advance()
        return hook

    # ------------------------------------------------------------------ staged overlap
    def set_staged(self, on: bool) -> None:
        """Staged overlap: the hooks only mark buckets ready; at the model's FLUSH POINTS (ops/_native.py: the gradient of the diffuser
        embedding, the gradient of every ConvNeXt stage input) the parked weight-gradient work is flushed in whole per-stage batches,
        every complete bucket is gathered and its all-reduce forked onto the side stream, where it runs beside the rest of the
        backward pass.  ~95 % of the gradient bytes are deferred weight gradients, so gathering from the per-leaf hooks instead (the
        plain ``overlap`` mode) has to flush early and chops those batches into per-bucket pieces (measured slower)."""
        from ..ops import _native
        self.staged = bool(on)
        _native.FLUSH_POINT[0] = self.flush_point if on else None

    def flush_point(self) -> None:
        if not self.staged or self._next >= len(self.buckets):
            return
        if not self.buckets[self._next]["ready"]:
            return                                 # nothing complete yet: keep the parked work for a bigger batch
        self._advance()                            # _gather flushes first

    def _advance(self) -> None:
        """Collectives are issued in BUCKET ORDER on every rank: bucket i goes out only once buckets < i have.  A bucket whose
        last gradient never arrives on some rank (a parameter unused in that step) is left to finish(), which continues in the
        same order, so the sequence of all-reduce sizes is identical on all ranks whatever the hook timing was."""
        while self._next < len(self.buckets) and self.buckets[self._next]["ready"]:
            b = self.buckets[self._next]
            self._gather(b)
            self._launch(b)
            self._next += 1

    def _pad(self, bucket, n: int, dtype):
        key = (n, dtype)
        z = bucket["pads"].get(key)
        if z is None:
            z = bucket["pads"][key] = torch.zeros(n, dtype=dtype, device=bucket["flat"].device)
        return z

    @torch.no_grad()
    def _gather(self, bucket) -> None:
        """Leaf gradients -> flat fp32 bucket: one batched concat per dtype segment (+ one cast for the
        low-precision segment) instead of a copy kernel per parameter; then restore the master .grad views."""
        self._flush_deferred()          # parked column reductions write their gradients now (no-op when none are pending)
        flat, k, nw = bucket["flat"], bucket["k_work"], bucket["n_work"]
        leaves, gviews, nhwc = bucket["leaves"], bucket["gviews"], bucket["nhwc"]
        sizes, padded = bucket["sizes"], bucket["padded"]

        def flat1d(g, cl):   # the gradient in the bucket's storage order (a view when its strides already match)
            return g.permute(0, 2, 3, 1).reshape(-1) if cl else g.reshape(-1)

        missing = []
        g16 = bucket["g16"]

        def seg(lo, hi, out, base):
            """leaves[lo:hi] -> out (flat fp32 slice, or the 16-bit communication buffer of the working-copy segment);
            ``base`` = offset of ``out`` inside the bucket."""
            if lo == hi:
                return
            gs = [l.grad for l in leaves[lo:hi]]
            offs = bucket["offsets"]
            if any(g is None for g in gs):   # a parameter unused this step: per-tensor copies; its slot is zero (world 1: FlatAdamW skips it like torch.optim.AdamW skips grad None)
                for i, (g, c) in enumerate(zip(gs, nhwc[lo:hi])):
                    dst = out[offs[lo + i] - base: offs[lo + i] - base + sizes[lo + i]]
                    if g is None:
                        dst.zero_()
                        missing.append(lo + i)
                    else:
                        dst.copy_(flat1d(g, c))
                return
            if flat.is_cuda:
                # ONE launch per 128 tensors and dtype: 16-bit gradient -> bucket slot directly (no concat + cast), table by value
                # in the kernel arguments (hipGraph-safe; torch.cat on ROCm is not, csrc/multicopy.hip).  Padding slots are never
                # written and stay zero.
                from .. import _lib as L
                by_dtype = {}
                for i, (g, c) in enumerate(zip(gs, nhwc[lo:hi])):
                    v = flat1d(g, c)
                    by_dtype.setdefault(v.dtype, ([], []))
                    by_dtype[v.dtype][0].append(v if v.is_contiguous() else v.contiguous())
                    by_dtype[v.dtype][1].append(offs[lo + i] - base)
                for ts, os_ in by_dtype.values():
                    L.multi_copy(ts, os_, out)
                return
            parts = []
            for i, (g, c) in enumerate(zip(gs, nhwc[lo:hi])):
                parts.append(flat1d(g, c))
                if padded[lo + i] != sizes[lo + i]:
                    parts.append(self._pad(bucket, padded[lo + i] - sizes[lo + i], g.dtype))
            if gs[0].dtype == out.dtype and all(g.dtype == out.dtype for g in gs):
                torch.cat(parts, out=out)
            else:
                out.copy_(torch.cat(parts))

        seg(0, k, g16 if g16 is not None else flat[:nw], 0)
        seg(k, len(leaves), flat[nw:], nw)
        bucket["missing"] = tuple(missing)
        for p, leaf, gv in zip(bucket["masters"], leaves, gviews):
            leaf.grad = None            # free the per-leaf gradient
            p.grad = gv
        bucket["done"] = True

    def _launch(self, bucket) -> None:
        if self.world == 1 and not self._force:
            return
        flat, g16, nw = bucket["flat"], bucket["g16"], bucket["n_work"]
        # RCCL averages inside the collective (ReduceOp.AVG); gloo (CPU tests) has no AVG: scale, then sum
        avg = dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        # payload: [16-bit working-copy segment] + [fp32 rest], or the whole fp32 bucket
        parts = [flat] if g16 is None else ([g16] + ([flat[nw:]] if flat.numel() > nw else []))

        def go():
            # The process group runs a collective on its OWN internal stream (forked from the stream that is current at the call); only
            # Work.wait() orders anything behind it.  The consumer joins per bucket (wait_bucket): the optimizer's stream waits for the
            # bucket's collectives and THEN widens the 16-bit payload into the fp32 bucket - never on this side stream, where nothing
            # would order the copy behind the all-reduce.
            for t in parts:
                if not avg:
                    t.div_(self.world)
                bucket["works"].append(dist.all_reduce(t, op=op, group=self.group, async_op=True))

        bucket["works"] = []
        if self._cuda and self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                go()
        else:
            go()

    @torch.no_grad()
    def wait_bucket(self, index: int, widen: bool = True) -> None:
        """Make the CURRENT stream wait for bucket ``index``'s all-reduce and widen its 16-bit payload into the fp32 bucket (no-op when
        nothing was launched for it).  With RCCL the wait is a stream-level event wait (capturable); with gloo the host blocks.
        ``widen=False``: the consumer reads the 16-bit payload itself (FlatAdamW, dgtd_adamw_flat_g16); the fp32 bucket's working-copy
        segment is then NOT refreshed for this step."""
        b = self.buckets[index]
        works = b.get("works")
        if not works:
            return
        for w in works:
            w.wait()
        # the Work objects stay alive until the end of the step (a Work that dies inside a hipGraph capture returns its completion
        # event to the process group's cache, where the next collective of the same capture would re-record it)
        self._works.extend(works)
        b["works"] = []
        if widen and b["g16"] is not None:
            b["flat"][:b["n_work"]].copy_(b["g16"])

    def finish(self) -> None:
        """Call after backward(): gathers/launches whatever the hooks did not (in bucket order), then fences the compute stream."""
        for b in self.buckets[self._next:]:
            self._gather(b)
            self._launch(b)
        self._next = len(self.buckets)
        self._defer(False)
        for b in self.buckets:
            self.wait_bucket(b["index"])
        self._works.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    @property
    def payload_bytes(self) -> int:
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)
