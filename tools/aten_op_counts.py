#!/usr/bin/env python
"""Host-side op inventory of one training step: every aten / dgtd op by call count and host self time, and the Python
call sites of the most frequent small ops (copies, adds, fills) — the launch-count budget of a host-bound step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = dgtd.nn.cod(compute_dtype=torch.bfloat16).to(dev).train()
red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
opt = dgtd.runner.build_optimizer(net)
data = dgtd.runner.SyntheticRGBD(512, 8, device=dev)
b = data.batch_at(0)


def step():
    red.zero_grad()
    loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
    loss.backward()
    red.finish()
    opt.step()
    red.refresh_working()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print(f"{'count':>6} {'self_us':>9} {'total_us':>9}  op")
for e in rows[:70]:
    print(f"{e.count:6d} {e.self_cpu_time_total:9.0f} {e.cpu_time_total:9.0f}  {e.key[:90]}")
want = sys.argv[1:] or ["aten::_to_copy", "aten::clone", "aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::cat",
                        "aten::sum", "aten::mul", "aten::contiguous"]
ks = prof.key_averages(group_by_input_shape=True)
for w in want:
    sel = sorted([e for e in ks if e.key == w], key=lambda e: -e.count)[:22]
    print(f"\n=== {w}: by input shapes")
    for e in sel:
        print(f"  {e.count:5d}  {str(e.input_shapes)[:150]}")
