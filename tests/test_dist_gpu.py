"""GPU, 2 processes sharing the one MI355X over gloo: exercises the CUDA side of GradReducer that the CPU test cannot —
post-accumulate hooks on the autograd thread, bucket gather, the side HIP stream with event fences, working-weight refresh —
and checks both ranks end up with identical, averaged gradients and identical weights after an optimizer step.
(RCCL itself needs one GPU per rank, which only the driver's multi-GPU node has.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import dgtd
        torch.cuda.set_device(0)
        dgtd.dist.init_process_group("gloo")
        torch.manual_seed(rank)                     # replicas start different; broadcast must fix that
        net = torch.nn.Sequential(dgtd.nn.Linear(64, 256), dgtd.nn.LayerNorm(256, eps=1e-6), torch.nn.GELU(),
                                  dgtd.nn.Linear(256, 64), dgtd.nn.LayerNorm(64, eps=1e-6), dgtd.nn.Linear(64, 1)).cuda()
        dgtd.dist.broadcast_parameters(net)
        red = dgtd.dist.GradReducer(net, bucket_bytes=32 << 10, exclude_prefixes=(), working_dtype=torch.bfloat16)
        assert len(red.buckets) >= 2 and red.comm_stream is not None
        opt = dgtd.runner.build_optimizer(net, lr=1e-2, custom_keys={})
        g = torch.Generator().manual_seed(7)
        x = torch.randn(2, 128, 64, generator=g)[rank].cuda()
        y = torch.randn(2, 128, 1, generator=g)[rank].cuda()
        for _ in range(3):
            red.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            ((out.float() - y) ** 2).mean().backward()
            red.finish()
            grads = [p.grad.detach().float().cpu().numpy().copy() for p in net.parameters()]
            opt.step()
            red.refresh_working()
        torch.cuda.synchronize()
        weights = [p.detach().float().cpu().numpy().copy() for p in net.parameters()]
        q.put((rank, grads, weights))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc(), None))
        raise


def test_reducer_streams_and_hooks_two_ranks_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, w = q.get(timeout=300)
        assert not isinstance(g, str), f"rank {r}: {g}"
        res[r] = (g, w)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b) and np.isfinite(a).all() and np.abs(a).sum() > 0   # same averaged gradient on both ranks
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)                                                      # replicas stay in lock-step
