"""Bucketed gradient all-reduce for the single-node data-parallel training step.

The reference gets this from mmengine's MMDistributedDataParallel with find_unused_parameters=True
(config/sod.yml:11): a per-step graph walk plus a per-step BN-buffer broadcast.  Here:
  * gradients live in flat per-bucket buffers (``param.grad`` is a view), so a bucket is reduced in place
    with ONE collective and no gather/scatter copies;
  * buckets follow reverse registration order (≈ gradient-readiness: Hitnet heads -> PVT stage 4..1 ->
    prompt decoders -> ConvNeXt 3..0 -> diffuser 1x1s), and each is launched from an autograd
    post-accumulate hook on a side HIP stream fenced by events, so RCCL traffic over xGMI overlaps the rest
    of the backward;
  * the parameters that never receive a gradient (5 tensors) are excluded statically instead of
    being discovered every step;
  * BatchNorm statistics stay per-rank (the reference uses plain BatchNorm2d, cod.py:362) and are not broadcast.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

# constructed but never called in the reference forward (cod.py:703-704, :1251): no gradient, ever
STATIC_UNUSED = ("hitnet.backbone.prompt_encoder.adaptor.", "hitnet.ca.", "hitnet.sa.")


def init_process_group(backend: Optional[str] = None) -> tuple:
    """One process per GPU; rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
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
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradReducer:
    """Owns flat gradient buckets for ``module`` and averages them across the process group."""

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 64 << 20, group=None,
                 exclude_prefixes: Sequence[str] = STATIC_UNUSED, overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        seen, params = set(), []
        for n, p in named:  # named_parameters() already de-duplicates the shared PReLU
            if id(p) in seen or any(n.startswith(e) for e in exclude_prefixes):
                continue
            seen.add(id(p))
            params.append((n, p))
        params.reverse()  # readiness order ≈ reverse registration order
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for n, p in params:
            cur.append((n, p))
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._seal(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._seal(cur)
        self.overlap = overlap and self.world > 1
        self._cuda = bool(self.buckets) and self.buckets[0]["flat"].is_cuda
        self.comm_stream = torch.cuda.Stream() if (self._cuda and self.overlap) else None
        self._works = []
        if self.overlap:
            for b in self.buckets:
                for _, p in b["params"]:
                    p.register_post_accumulate_grad_hook(self._make_hook(b))

    def _seal(self, items) -> None:
        p0 = items[0][1]
        total = sum(p.numel() for _, p in items)
        flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        off = 0
        for _, p in items:
            p.grad = flat[off:off + p.numel()].view_as(p)  # gradient-as-bucket-view
            off += p.numel()
        self.buckets.append({"flat": flat, "params": items, "pending": len(items), "n": len(items)})

    def _make_hook(self, bucket):
        def hook(_param):
            if not self.overlap or torch.cuda.is_current_stream_capturing():
                return
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                self._launch(bucket)
        return hook

    def _launch(self, bucket) -> None:
        flat = bucket["flat"]
        if self._cuda and self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                flat.div_(self.world)
                w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat.div_(self.world)
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)

    def zero_grad(self) -> None:
        """Replaces optimizer.zero_grad(): one memset per bucket, grads stay views of the flat buffers."""
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = b["n"]
            off = 0
            for _, p in b["params"]:
                if p.grad is None or p.grad.data_ptr() != b["flat"].data_ptr() + off * b["flat"].element_size():
                    p.grad = b["flat"][off:off + p.numel()].view_as(p)
                off += p.numel()

    def finish(self) -> None:
        """Call after backward(): launches anything not launched by hooks and fences the compute stream."""
        if self.world == 1:
            return
        if not self.overlap:
            for b in self.buckets:
                self._launch(b)
        elif False:
            pass
        else:
            for b in self.buckets:  # a bucket whose params did not all fire (should not happen) is still reduced
                if b["pending"] != 0:
                    self._launch(b)
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    @property
    def payload_bytes(self) -> int:
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)
