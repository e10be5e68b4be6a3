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


# ---------------------------------------------------------------------------------------------- the real model, world 2
def _cod_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        torch.set_num_threads(3)
        import dgtd
        from oracle import cod_cpu, filler
        dgtd.dist.init_process_group("gloo")
        S = 32
        net = cod_cpu.cod(S).train()                     # the reference architecture (BatchNorm in train mode, shared PReLU, unused heads)
        filler.fill_module(net)
        if rank == 1:
            with torch.no_grad():
                net.hitnet.out_CFM.bias.add_(1.0)        # replicas start different: the broadcast must fix it
        dgtd.dist.broadcast_parameters(net)
        red = dgtd.dist.GradReducer(net, bucket_bytes=32 << 20)
        names = [n for b in red.buckets for n in b["names"]]
        x, d, l = filler.synthetic_batch(4, S, seed=21)
        sampler = dgtd.runner.DefaultSampler(4, shuffle=True, seed=5, rank=rank, world=world)
        idx = list(sampler)
        red.zero_grad()
        loss = net(None, x[idx], l[idx], list(d[idx]), mode="loss")["loss"]
        loss.backward()
        red.finish()
        probe = {n: p.grad.detach().flatten()[:64].clone().numpy() for n, p in net.named_parameters() if p.grad is not None and p.requires_grad}
        norms = {n: float(p.grad.double().norm()) for n, p in net.named_parameters() if p.grad is not None}
        bn = net.hitnet.conv4.bn.running_mean.detach().clone().numpy()
        q.put((rank, {"idx": idx, "names": names, "probe": probe, "norms": norms, "bn": bn, "n_buckets": len(red.buckets),
                      "loss": float(loss.detach())}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))
        raise


def test_real_model_world2_equals_mean_of_per_rank_losses():
    """with per-rank BatchNorm statistics (plain BatchNorm2d, cod.py:362, never synchronised), the all-reduced
    gradient of 2 ranks x 2 samples equals the gradient of the MEAN of the two per-rank losses computed in one process, each half
    normalised with its own batch statistics.  Runs the reference architecture (oracle restatement: 7 BN layers in train mode, the
    ONE PReLU shared by 8 CABs x 4 iterations, the 5 never-used tensors) through dist.GradReducer over gloo with the DefaultSampler
    split."""
    from oracle import cod_cpu, filler
    import dgtd
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cod_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for r, v in got.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = got[0], got[1]
    assert sorted(a["idx"] + b["idx"]) == [0, 1, 2, 3] and a["n_buckets"] >= 5
    # the statically excluded tensors are in no bucket; the shared PReLU is there exactly once
    assert not any(n.startswith(dgtd.dist.reducer.STATIC_UNUSED) for n in a["names"])
    assert sum("body.1.weight" in n for n in a["names"]) == 1
    assert len(a["names"]) == 846
    for n in a["probe"]:                                  # ranks agree bit for bit
        assert (a["probe"][n] == b["probe"][n]).all(), n
    assert not (a["bn"] == b["bn"]).all()                 # BN statistics stay per rank
    # single process: mean of the two per-rank losses, each half through its own BatchNorm batch statistics
    S = 32
    net = cod_cpu.cod(S).train()
    filler.fill_module(net)
    x, d, l = filler.synthetic_batch(4, S, seed=21)
    total = 0.0
    for idx in (a["idx"], b["idx"]):
        total = total + net(None, x[idx], l[idx], list(d[idx]), mode="loss")["loss"] / 2
    total.backward()
    assert abs(float(total.detach()) - (a["loss"] + b["loss"]) / 2) < 1e-5
    bad = []
    for n, p in net.named_parameters():
        if p.grad is None:
            assert n.startswith(dgtd.dist.reducer.STATIC_UNUSED), n
            continue
        want = float(p.grad.double().norm())
        if abs(a["norms"][n] - want) > 2e-3 * want + 1e-7:   # fp32 re-association (3 worker threads per rank vs all cores here)
            bad.append((n, a["norms"][n], want))
        probe = p.grad.detach().flatten()[:64]
        torch.testing.assert_close(torch.from_numpy(a["probe"][n]), probe, rtol=5e-3, atol=1e-6 + 2e-3 * float(probe.abs().max()), msg=lambda m: f"{n}: {m}")
    assert not bad, bad[:5]
