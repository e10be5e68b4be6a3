#!/usr/bin/env python
"""cProfile of the host side of the training step (which Python / dispatch paths cost the enqueue time)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgtd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = dgtd.nn.cod(compute_dtype=torch.bfloat16).to(dev).train()
red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
opt = dgtd.runner.build_optimizer(net)
data = dgtd.runner.SyntheticRGBD(int(os.environ.get("SIZE", 512)), 8, device=dev)
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
t = time.perf_counter()
for _ in range(3):
    step()
host = time.perf_counter() - t
torch.cuda.synchronize()
print(f"host enqueue {host / 3 * 1e3:.1f} ms/step, wall {(time.perf_counter() - t) / 3 * 1e3:.1f} ms/step")
# phase split
for name, fn in [("zero_grad", red.zero_grad)]:
    t = time.perf_counter(); fn(); print(name, (time.perf_counter() - t) * 1e3, "ms")
t = time.perf_counter(); loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]; t1 = time.perf_counter()
loss.backward(); t2 = time.perf_counter(); red.finish(); t3 = time.perf_counter(); opt.step(); t4 = time.perf_counter(); red.refresh_working(); t5 = time.perf_counter()
print(f"forward {1e3*(t1-t):.1f}  backward {1e3*(t2-t1):.1f}  finish {1e3*(t3-t2):.1f}  opt {1e3*(t4-t3):.1f}  refresh {1e3*(t5-t4):.1f} ms (host)")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
