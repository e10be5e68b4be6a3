#!/usr/bin/env python
"""Where do the per-step dtype casts / layout copies come from?  A TorchDispatchMode logs every aten._to_copy / clone /
copy_ with shape, dtypes, the innermost package frame (forward) or the autograd node (backward)."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = dgtd.nn.cod(compute_dtype=torch.bfloat16).to(dev).train()
red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
opt = dgtd.runner.build_optimizer(net)
data = dgtd.runner.SyntheticRGBD(512, 8, device=dev)
b = data.batch_at(0)
WATCH = {"aten._to_copy.default", "aten.clone.default", "aten.copy_.default"}
sites = collections.Counter()


def site():
    node = torch._C._current_autograd_node()
    if node is not None:
        return "bwd:" + type(node).__name__
    for fr in reversed(traceback.extract_stack()):
        if ("depth-guided" in fr.filename or "dgtd" in fr.filename) and "tools/" not in fr.filename:
            return f"fwd:{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name in WATCH:
            src = args[1] if name == "aten.copy_.default" else args[0]
            dst = out
            lay = "" if src.stride() == dst.stride() or src.numel() != dst.numel() else " relayout"
            sites[(name.split(".")[1], tuple(src.shape), f"{str(src.dtype)[6:]}->{str(dst.dtype)[6:]}{lay}", site())] += 1
        return out


def step():
    red.zero_grad()
    loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
    loss.backward()
    red.finish()
    opt.step()
    red.refresh_working()


for _ in range(2):
    step()
with Log():
    step()
torch.cuda.synchronize()
print(f"{sum(sites.values())} copies/casts in one step")
for k, n in sites.most_common(90):
    print(f"{n:5d}  {k[0]:9s} {str(k[1]):24s} {k[2]:28s} {k[3]}")
