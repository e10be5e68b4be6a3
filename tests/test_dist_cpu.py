"""CPU, world_size 2 over gloo: the bucketed gradient reducer (dgtd.dist.GradReducer) gives every rank the average of
the per-rank gradients == the gradient of the mean loss over the concatenated batch (SURVEY §8(e) equivalence test),
with and without low-precision working weights; plus the lr-multiplier rule of the optimizer builder."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy():
    import dgtd
    torch.manual_seed(0)
    return torch.nn.Sequential(dgtd.nn.Linear(16, 32), torch.nn.LayerNorm(32), torch.nn.GELU(), dgtd.nn.Linear(32, 8),
                               torch.nn.PReLU(), dgtd.nn.Linear(8, 1))


def _worker(rank, world, port, working, q):
    try:
        _worker_body(rank, world, port, working, q)
    except Exception:  # surface the rank's traceback in the parent instead of a bare exit code
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))
        raise


def _worker_body(rank, world, port, working, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import dgtd
    dgtd.dist.init_process_group("gloo")
    net = _toy()
    if rank == 1:  # replicas start different: broadcast must fix it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    dgtd.dist.broadcast_parameters(net)
    red = dgtd.dist.GradReducer(net, bucket_bytes=1 << 10, exclude_prefixes=(), working_dtype=torch.bfloat16 if working else None)
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 1, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):  # two steps: the second checks zero_grad()/re-gather
        red.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=working):
            out = net(xs)
        loss = ((out.float() - ys) ** 2).mean()
        loss.backward()
        red.finish()
    grads = {n: p.grad.detach().numpy().copy() for n, p in net.named_parameters()}   # plain bytes: no fd passing between processes
    q.put((rank, grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("working", [False, True])
def test_grad_reducer_matches_large_batch(working):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, working, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for r, v in got.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _toy()
    g = torch.Generator().manual_seed(100)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    tol = dict(rtol=5e-2, atol=5e-3) if working else dict(rtol=1e-5, atol=1e-6)
    for n, p in net.named_parameters():
        a, b = torch.from_numpy(got[0][n]), torch.from_numpy(got[1][n])
        torch.testing.assert_close(a, b, rtol=0, atol=0)   # ranks agree bit-for-bit
        torch.testing.assert_close(a, p.grad, **tol)


def test_single_process_reducer_and_optimizer_groups():
    import dgtd
    net = _toy()
    red = dgtd.dist.GradReducer(net, exclude_prefixes=(), working_dtype=torch.bfloat16)
    opt = dgtd.runner.build_optimizer(net, fused=False)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(list(net.parameters()))
    before = [p.detach().clone() for p in net.parameters()]
    red.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = net(torch.randn(4, 16))
    out.float().sum().backward()
    red.finish()
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in net.parameters())
    opt.step()
    red.refresh_working()
    assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters()))
    lin = net[0]
    torch.testing.assert_close(lin._w.float(), lin.weight.detach().bfloat16().float())  # working copy follows the master
    keys = dgtd.runner.optim.SOD_CUSTOM_KEYS
    assert dgtd.runner.lr_mult_for("hitnet.decoder_level1.0.body.0.weight", keys) == 1.0
    assert dgtd.runner.lr_mult_for("hitnet.backbone.block1.0.attn.q.weight", keys) == 0.2
    assert dgtd.runner.lr_mult_for("hitnet.backbone.prompt_encoder.encoder2.stages.2.5.gamma", keys) == 0.02
    assert dgtd.runner.lr_mult_for("hitnet.backbone.prompt_encoder.encoder2.convs.0.weight", keys) == 0.2
