#!/usr/bin/env python
"""Experiment: capture zero_grad+forward+loss+backward in ONE hipGraph (rocFFT, RCCL and AdamW stay outside) and compare
host/wall time per step with eager."""
import os
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
data = dgtd.runner.SyntheticRGBD(512, 8, device=dev)
batches = [data.batch_at(i) for i in range(2)]
static = {k: torch.stack(batches[0][k]).clone() for k in ("input", "label", "depth")}
static["x_hp"] = net.high_pass(static["input"]).clone()


def fwd_bwd():
    red.zero_grad()
    loss = net(None, static["input"], static["label"], static["depth"], mode="loss", x_hp=static["x_hp"])["loss"]
    loss.backward()
    return loss


def timeit(fn, n=8):
    torch.cuda.synchronize()
    t = time.perf_counter(); host = 0.0
    for i in range(n):
        th = time.perf_counter(); fn(i); host += time.perf_counter() - th
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, host / n * 1e3


def eager(i):
    b = batches[i % 2]
    for k in ("input", "label", "depth"):
        static[k].copy_(torch.stack(b[k]))
    static["x_hp"].copy_(net.high_pass(static["input"]))
    loss = fwd_bwd(); red.finish(); opt.step(); red.refresh_working()
    return loss


for i in range(3):
    l = eager(i)
print("eager loss", l.item(), " wall/host ms:", timeit(eager))

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
red.freeze_for_graph()
print(f"capture took {time.perf_counter() - t0:.1f} s")


def graphed(i):
    b = batches[i % 2]
    for k in ("input", "label", "depth"):
        static[k].copy_(torch.stack(b[k]))
    static["x_hp"].copy_(net.high_pass(static["input"]))
    g.replay()
    red.zero_grad(); red.finish(); opt.step(); red.refresh_working()
    return static_loss


for i in range(3):
    l = graphed(i)
print("graph loss", l.item(), " wall/host ms:", timeit(graphed))
tr = time.perf_counter(); g.replay(); th = time.perf_counter() - tr; torch.cuda.synchronize()
print(f"one replay: host {th * 1e3:.1f} ms, total {(time.perf_counter() - tr) * 1e3:.1f} ms")
