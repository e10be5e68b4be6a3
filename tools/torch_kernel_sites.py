#!/usr/bin/env python
"""Which torch (non-dgtd) device ops run in one training step, from where?  A TorchDispatchMode logs every aten op that is not a
pure view, with its first tensor shape/dtype and the call site (forward: innermost package frame; backward: autograd node)."""
import collections
import os
import sys
import traceback

os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = dgtd.nn.cod(compute_dtype=torch.bfloat16).to(dev).train()
red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
opt = dgtd.runner.FlatAdamW(red)
data = dgtd.runner.SyntheticRGBD(512, 8, device=dev)
b = data.batch_at(0)
VIEWS = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "narrow", "as_strided", "t.default",
         "detach", "alias", "unbind", "split", "_unsafe_view", "unfold", "lift_fresh", "empty", "new_empty", "sym_", "stride", "size",
         "is_", "_local_scalar", "prim", "result_type", "_has_compatible", "chunk")
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
        if any(v in name for v in VIEWS):
            return out
        t = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if t is None and args and isinstance(args[0], (list, tuple)):
            t = next((a for a in args[0] if isinstance(a, torch.Tensor)), None)
        if t is not None and not t.is_cuda:
            return out
        shp = tuple(t.shape) if t is not None else ()
        sites[(name.replace("aten.", ""), shp, str(t.dtype)[6:] if t is not None else "", site())] += 1
        return out


def step():
    red.zero_grad()
    loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
    loss.backward()
    red.finish()
    opt.step()


for _ in range(2):
    step()
with Log():
    step()
torch.cuda.synchronize()
print(f"{sum(sites.values())} non-view aten/dgtd ops in one step")
by_op = collections.Counter()
for k, n in sites.items():
    by_op[k[0]] += n
print("by op:", by_op.most_common(60))
for k, n in sites.most_common(200):
    print(f"{n:5d}  {k[0]:34s} {str(k[1]):26s} {k[2]:9s} {k[3]}")
