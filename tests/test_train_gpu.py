"""GPU: the training loop around the model - config-driven Runner (train / val / checkpoint / resume), the flat AdamW with the
reference's AMP recipe (fp16 + fp32 masters + dynamic loss scaling, config/sod.yml:57) and the fp16 vs bf16 loss curves."""
import math
import os

import pytest
import torch

from oracle import filler

pytestmark = pytest.mark.gpu

# config/sod.yml with the run shortened (3 epochs) and the size-dependent entries adapted to the synthetic dataset; every key the
# runner reads keeps the reference's spelling (config/sod.yml:1-104)
SOD_SHORT = """
train_cfg:
  by_epoch: &by_epoch True
  max_epochs: &max_epochs 3
  val_interval: 3
val_cfg: {}
find_unused_parameters: True
train_dataloader:
  batch_size: 2
  num_workers: 8
  sampler:
    type: DefaultSampler
    shuffle: True
val_dataloader:
  batch_size: 1
  num_workers: 8
  sampler:
    type: DefaultSampler
    shuffle: False
model:
  type: cod
  win_size: 22
  filter_ratio: 0.9
  using_depth: True
  using_sam: True
  drop_path_rate: 0.0
optim_wrapper:
  type: AmpOptimWrapper
  optimizer:
    type: AdamW
    lr: 0.0005
    weight_decay: 0.1
  paramwise_cfg:
    bypass_duplicate: True
    custom_keys:
      hitnet.backbone:
        lr_mult: 0.2
      hitnet.backbone.prompt_encoder.encoder2.downsample_layers:
        lr_mult: 0.02
      hitnet.backbone.prompt_encoder.encoder2.stages.0:
        lr_mult: 0.02
      hitnet.backbone.prompt_encoder.encoder2.stages.1:
        lr_mult: 0.02
      hitnet.backbone.prompt_encoder.encoder2.stages.2:
        lr_mult: 0.02
      hitnet.backbone.prompt_encoder.encoder2.stages.3:
        lr_mult: 0.02
param_scheduler:
  type: CosineAnnealingLR
  by_epoch: *by_epoch
  T_max: *max_epochs
val_evaluator:
  - type: Emeasure
  - type: Fmeasure
  - type: Smeasure
  - type: MAE
default_hooks:
  logger:
    type: LoggerHook
    interval: 5
  checkpoint:
    type: CheckpointHook
    by_epoch: *by_epoch
    interval: 1
custom_hooks:
  -
    type: our_init
"""


@pytest.fixture(scope="module")
def dgtd():
    import dgtd as m
    m._lib.load()
    return m


def _runner(dgtd, tmp, dtype=torch.bfloat16):
    cfg = dgtd.runner.load_config(SOD_SHORT)
    torch.manual_seed(0)
    logs = []
    r = dgtd.runner.Runner(cfg, device="cuda", compute_dtype=dtype, work_dir=str(tmp), log=logs.append, seed=3)
    filler.fill_module(r.model)            # identical starting weights for every runner of this module (load hooks refresh the working copies)
    return r, logs


def test_runner_trains_validates_checkpoints_and_resumes(dgtd, tmp_path):
    """Runner.train from the YAML text: 3 epochs x 3 iterations (6 samples, batch 2, DefaultSampler shuffle), MAE evaluator on the
    val split every val_interval epochs, a checkpoint per epoch.  A second runner resumes from epoch_2.pth and must reproduce the
    third epoch of the uninterrupted run: same losses, same final weights (FlatAdamW moments + step, cosine schedule, epoch)."""
    S = 64
    train_ds = dgtd.runner.SyntheticRGBD(S, 2, device="cuda", length=6)
    val_ds = dgtd.runner.SyntheticRGBD(S, 1, device="cuda", seed=99, length=3)
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        # fp32 compute for the resume comparison: in the 16-bit modes fp32-atomic summation order (attention dK/dV, PReLU slope,
        # diffuser parameters) decides single bf16 roundings of the working copies, so two runs of the SAME step sequence drift apart
        # by ~1e-3 in the loss after a few updates; in fp32 the same effect stays at the 1e-6 level
        a, logs = _runner(dgtd, tmp_path / "a", torch.float32)
        assert isinstance(a.optimizer, dgtd.runner.FlatAdamW) and [type(e).__name__ for e in a.evaluators] == ["MAE"]
        assert sum("skipped" in m for m in logs if isinstance(m, str)) == 3          # E/F/S-measure: third-party, announced
        la = a.train(lambda e: a.loader(train_ds, "train", e), val_loader_fn=lambda: a.loader(val_ds, "val"))
        assert len(la) == 9 and all(math.isfinite(v) for v in la)
        assert all(os.path.exists(tmp_path / "a" / f"epoch_{e}.pth") for e in (1, 2, 3))
        val_lines = [m for m in logs if isinstance(m, str) and " val " in m]
        assert len(val_lines) == 1 and "MAE" in val_lines[0]
        assert abs(a.optimizer.param_groups[0]["lr"]) < 1e-12                         # cosine reached 0 after T_max epochs
        blob = torch.load(tmp_path / "a" / "epoch_2.pth", weights_only=False)
        assert set(blob) >= {"state_dict", "optimizer", "param_schedulers", "meta"} and blob["meta"]["epoch"] == 2
        assert len(blob["state_dict"]) == 879 and len(blob["optimizer"]["state"]) == 846
        k0 = blob["optimizer"]["param_names"].index("hitnet.backbone.block3.5.attn.kv.weight")
        assert tuple(blob["optimizer"]["state"][k0]["exp_avg"].shape) == (640, 320)   # torch.optim.AdamW layout: logical shapes
        b, _ = _runner(dgtd, tmp_path / "b", torch.float32)
        b.resume(str(tmp_path / "a" / "epoch_2.pth"))
        assert b.epoch == 2 and b.optimizer.steps == 6
        lb = b.train(lambda e: b.loader(train_ds, "train", e))
        assert len(lb) == 3
        for x, y in zip(la[6:], lb):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (la[6:], lb)
        for (k, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
            torch.testing.assert_close(p, q, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f"{k}: {m}")
        # validation alone (script/test.sh: `-m val`): eval mode, predict path, metrics dict
        m = b.validate(b.loader(val_ds, "val"))
        assert set(m) == {"MAE"} and 0.0 <= m["MAE"] <= 1.0
    finally:
        torch.backends.cudnn.deterministic = old


def test_runner_bf16_production_path(dgtd, tmp_path):
    """The configuration bench.py measures (bf16 working copies + FlatAdamW) through Runner.train: one epoch, finite decreasing-ish
    losses, working copies in step with the masters after the last update, checkpoint written."""
    train_ds = dgtd.runner.SyntheticRGBD(64, 2, device="cuda", length=6)
    r, _ = _runner(dgtd, tmp_path / "c", torch.bfloat16)
    losses = r.train(lambda e: r.loader(train_ds, "train", e), epochs=1)
    assert len(losses) == 3 and all(math.isfinite(v) for v in losses)
    q = r.model.hitnet.backbone.block1[0].attn.q
    torch.testing.assert_close(q._w.float(), q.weight.detach().bfloat16().float(), rtol=0, atol=0)
    assert os.path.exists(tmp_path / "c" / "epoch_1.pth")


def test_flat_adamw_state_loads_into_torch_adamw(dgtd):
    """the optimizer entry of a checkpoint is torch.optim.AdamW's state_dict layout, so torch (and mmengine) can read it."""
    import copy
    torch.manual_seed(0)
    net = torch.nn.Sequential(dgtd.nn.modules.Linear(24, 40), torch.nn.LayerNorm(40), dgtd.nn.modules.Conv2d(4, 6, 3, padding=1)).cuda()
    twin = copy.deepcopy(net)
    red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16, exclude_prefixes=())
    opt = dgtd.runner.FlatAdamW(red, lr=1e-2, custom_keys={})
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(3):
        for b in red.buckets:
            b["flat"].copy_(torch.randn(b["flat"].shape, device="cuda", generator=g) * 0.1)
        opt.step()
    sd = opt.state_dict()
    names = sd["param_names"]
    params = dict(twin.named_parameters())
    topt = torch.optim.AdamW([params[n] for n in names], lr=1e-2, weight_decay=0.1)
    topt.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    i = names.index("2.weight")
    assert tuple(topt.state[params["2.weight"]]["exp_avg"].shape) == (6, 4, 3, 3) and int(topt.state[params["2.weight"]]["step"]) == 3
    # element order: the flat bucket stores the 3x3 kernel O,H,W,I; the exported state is logical O,I,H,W
    b = red.buckets[0]
    j = b["names"].index("2.weight")
    raw = opt.state[0]["exp_avg"][b["offsets"][j]:b["offsets"][j] + b["sizes"][j]].view(6, 3, 3, 4).permute(0, 3, 1, 2)
    torch.testing.assert_close(sd["state"][i]["exp_avg"].cuda(), raw, rtol=0, atol=0)
    # and back: a fresh FlatAdamW restores from it
    red2 = dgtd.dist.GradReducer(twin, working_dtype=torch.bfloat16, exclude_prefixes=(), bucket_bytes=64)   # different bucketing on purpose
    opt2 = dgtd.runner.FlatAdamW(red2, lr=1e-2, custom_keys={})
    opt2.load_state_dict(sd)
    assert opt2.steps == 3
    sd2 = opt2.state_dict()
    for a_, b_ in zip(sd["state"].values(), [sd2["state"][sd2["param_names"].index(n)] for n in names]):
        torch.testing.assert_close(a_["exp_avg_sq"], b_["exp_avg_sq"], rtol=0, atol=0)


def test_loss_scaler_matches_torch_grad_scaler(dgtd):
    """The fused AMP step (found_inf pass + unscale inside dgtd_adamw_flat_amp + device-side scale update) against
    torch.amp.GradScaler + torch.optim.AdamW on identical scaled gradients: an overflowed step is skipped and halves the scale,
    `growth_interval` clean steps double it, skipped steps do not advance the bias corrections."""
    import copy
    torch.manual_seed(0)
    net = torch.nn.Sequential(dgtd.nn.modules.Linear(32, 48), torch.nn.LayerNorm(48), dgtd.nn.modules.Linear(48, 7)).cuda()
    twin = copy.deepcopy(net)
    red = dgtd.dist.GradReducer(net, working_dtype=torch.float16, exclude_prefixes=())
    scaler = dgtd.runner.LossScaler("cuda", init_scale=1024.0, growth_interval=3)
    opt = dgtd.runner.FlatAdamW(red, lr=1e-2, weight_decay=0.1, custom_keys={}, scaler=scaler)
    assert red.buckets[0]["wflat"].dtype == torch.float16
    tparams = dict(twin.named_parameters())
    order = [n for b in red.buckets for n in b["names"]]
    topt = torch.optim.AdamW([tparams[n] for n in order], lr=1e-2, weight_decay=0.1)
    ts = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
    ts.scale(torch.zeros(1, device="cuda"))                 # GradScaler creates its device state lazily
    g = torch.Generator(device="cuda").manual_seed(2)
    overflow_steps = {1, 5}
    for step in range(9):
        scale_now = scaler.get_scale()
        assert scale_now == ts.get_scale(), (step, scale_now, ts.get_scale())
        for b in red.buckets:
            true_grad = torch.randn(b["flat"].shape, device="cuda", generator=g) * 0.1
            scaled = true_grad * scale_now
            if step in overflow_steps:
                scaled[b["offsets"][0] + 1] = float("inf") if step == 1 else float("nan")
            b["flat"].copy_(scaled)
            for n, off, size, shape in zip(b["names"], b["offsets"], b["sizes"], b["shapes"]):
                tparams[n].grad = scaled[off:off + size].view(shape).clone()
        opt.step()
        ts.step(topt)        # unscale_ + skip on inf/nan
        ts.update()
    assert scaler.get_scale() == ts.get_scale()
    assert opt.steps == 9 - len(overflow_steps)
    for n, p in net.named_parameters():
        torch.testing.assert_close(p, tparams[n], rtol=2e-5, atol=2e-6, msg=lambda m, n=n: f"{n}: {m}")
    lin = net[0]
    torch.testing.assert_close(lin._w.float(), lin.weight.detach().half().float(), rtol=0, atol=0)   # fp16 working copy rewritten in the same pass
    sd = opt.state_dict()
    assert sd["loss_scaler"]["scale"] == ts.get_scale() and int(sd["state"][0]["step"]) == 7


@pytest.mark.parametrize("S,B", [(64, 2)])
def test_fp16_training_tracks_bf16_loss_curve(dgtd, S, B):
    """`compute_dtype=torch.float16` + LossScaler over 20 synthetic steps follows the bf16 run (same weights,
    same batches, DropPath off): finite losses, no divergence, the curves within a few percent of each other, and the scaler
    ends on a finite scale having taken (almost) every step."""
    def run(dtype):
        torch.manual_seed(0)
        net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=dtype)
        filler.fill_module(net)
        net = net.cuda().train()
        red = dgtd.dist.GradReducer(net, working_dtype=dtype)
        scaler = dgtd.runner.LossScaler("cuda") if dtype == torch.float16 else None
        opt = dgtd.runner.FlatAdamW(red, lr=1e-4, scaler=scaler)
        data = dgtd.runner.SyntheticRGBD(S, B, device="cuda")
        losses = []
        for i in range(20):
            b = data.batch_at(i % 4)
            red.zero_grad()
            loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
            (scaler.scale(loss) if scaler else loss).backward()
            red.finish()
            opt.step()
            losses.append(loss.item())
        return losses, opt, scaler
    l16, opt16, scaler = run(torch.float16)
    lbf, _, _ = run(torch.bfloat16)
    assert all(math.isfinite(v) for v in l16 + lbf), (l16, lbf)
    assert opt16.steps >= 15, opt16.steps                      # the first steps may overflow at scale 65536 and are skipped
    assert math.isfinite(scaler.get_scale()) and scaler.get_scale() >= 1.0
    # both runs learn (4 repeating batches), and the curves stay together
    assert sum(l16[-4:]) < sum(l16[:4]) and sum(lbf[-4:]) < sum(lbf[:4]), (l16, lbf)
    assert abs(l16[0] - lbf[0]) < 0.02 * abs(lbf[0]), (l16[0], lbf[0])         # same weights, same batch: only the arithmetic differs
    m16, mbf = sum(l16[-4:]) / 4, sum(lbf[-4:]) / 4                           # fp16 lags by the few steps its scaler skipped
    assert abs(m16 - mbf) < 0.10 * abs(mbf), (l16, lbf)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=str)
def test_graphed_step_matches_eager(dgtd, dtype):
    """the whole training step (forward + loss + backward + bucket gather + AdamW) captured as ONE hipGraph and
    replayed must follow the eager step: same losses and weights over 3 steps, the capture's eager warm-up steps leaving no trace
    (parameters, moments, step counter, BatchNorm statistics are restored: ADVICE r2) (fp32: to fp32-atomic noise;
    bf16: finite and close), with the learning rate changed between replays (device-resident lr) and the optimizer's step count
    advancing on the device."""
    S, B, W = 64, 2, 2
    data = dgtd.runner.SyntheticRGBD(S, B, device="cuda")
    batches = [data.batch_at(i) for i in range(3)]

    def make():
        torch.manual_seed(0)
        net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=dtype)
        filler.fill_module(net)
        net = net.cuda().train()
        red = dgtd.dist.GradReducer(net, working_dtype=dtype)
        opt = dgtd.runner.FlatAdamW(red, lr=1e-4, graph_safe=True)
        return net, red, opt

    def eager_step(net, red, opt, b):
        red.zero_grad()
        loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
        loss.backward()
        red.finish()
        opt.sync_lr()
        opt.step()
        return loss.item()

    net_e, red_e, opt_e = make()     # no warm-up on the eager side: the captured step restores weights / moments / statistics after its own
    want = []
    for i in range(3):
        if i == 2:
            for g in opt_e.param_groups:
                g["lr"] *= 0.5
        want.append(eager_step(net_e, red_e, opt_e, batches[i]))

    net_g, red_g, opt_g = make()
    stepper = dgtd.runner.GraphedTrainStep(net_g, red_g, opt_g, warmup=W)
    stepper.capture(batches[0])
    got = []
    for i in range(3):
        if i == 2:
            for g in opt_g.param_groups:
                g["lr"] *= 0.5
        got.append(stepper(batches[i]).item())
        # the round-1 NaN trigger: a host synchronisation followed by eager torch.cat / torch.stack between replays (their pinned
        # staging buffers used to be what the captured gather re-read on replay)
        torch.cuda.synchronize()
        junk = [torch.stack([torch.randn(3, 64, 64, device="cuda") for _ in range(8)]) for _ in range(4)]
        junk.append(torch.cat([torch.randn(100, device="cuda") for _ in range(200)]))
        torch.cuda.synchronize()
        del junk
    assert opt_g.steps == opt_e.steps == 3
    tol = 1e-4 if dtype == torch.float32 else 2e-2      # fp32: atomic summation order in a few backward kernels
    for a_, b_ in zip(got, want):
        assert math.isfinite(a_) and abs(a_ - b_) <= tol * max(1.0, abs(b_)), (got, want)
    if dtype == torch.float32:
        bad = 0
        for (k, p), (_, q) in zip(net_g.named_parameters(), net_e.named_parameters()):
            # AdamW turns gradient noise on elements with a tiny second moment into O(lr) differences: allow isolated elements
            bad += int(((p - q).abs() > 2e-5 + 1e-3 * q.abs()).sum())
            torch.testing.assert_close(p, q, rtol=1e-2, atol=5e-4, msg=lambda m, k=k: f"{k}: {m}")
        assert bad < 20000, bad                                  # of 114 M elements


def test_deferred_column_reductions_are_bit_identical(dgtd):
    """The batched second stage (dgtd_multi_reduce over every parked LayerNorm / bias / layer-scale reduction of a backward pass)
    runs the same sums in the same order as the per-op second stage: every parameter gradient of a bf16 training step must be
    bit-identical with and without deferral, and nothing may stay parked after finish()."""
    from dgtd.dist import reducer as R
    nat = dgtd.ops._native.ops()
    assert nat is not None
    S, B = 64, 2
    x, d, l = (t.cuda() for t in filler.synthetic_batch(B, S, seed=4))
    old_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    grads = {}
    try:
        for defer in (False, True):
            R.DEFER = defer
            net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.bfloat16)
            filler.fill_module(net)
            net = net.cuda().train()
            red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
            red.zero_grad()
            loss = net(None, x, l, d, mode="loss")["loss"]
            done = nat.flushed_reductions()
            loss.backward()
            # parked work is flushed by the engine callback at the end of the backward pass (and whatever a hook flushed earlier)
            assert nat.pending_reductions() == 0
            if defer:
                assert nat.flushed_reductions() - done > 200   # LayerNorms + Linear biases of both trunks were parked
            else:
                assert nat.flushed_reductions() == done
            red.finish()
            assert nat.pending_reductions() == 0
            grads[defer] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    finally:
        R.DEFER = True
        torch.backends.cudnn.deterministic = old_det
    # tensors fed by fp32 atomics (attention dK/dV, PReLU slope, diffuser parameters) differ run to run by themselves; everything the
    # deferred path touches directly (LayerNorm, biases, gamma) must be identical, and nothing else may move beyond atomic noise
    direct = [k for k in grads[True] if k.endswith(("norm.weight", "norm.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", ".gamma"))
              and "block" not in k.split("norm")[0][-8:]]
    assert len(direct) > 50
    same = sum(torch.equal(grads[True][k], grads[False][k]) for k in grads[True])
    for k in grads[True]:
        torch.testing.assert_close(grads[True][k], grads[False][k], rtol=2e-2, atol=1e-5, msg=lambda m, k=k: f"{k}: {m}")
    enc = [k for k in grads[True] if "encoder2.stages.3" in k and (k.endswith("norm.weight") or k.endswith("pwconv1.bias") or k.endswith(".gamma"))]
    assert enc and all(torch.equal(grads[True][k], grads[False][k]) for k in enc), "last ConvNeXt stage: no atomics upstream in the backward"
    assert same > len(grads[True]) // 4


def test_batched_weight_gradient_gemms_match_per_layer_path(dgtd):
    """Deferred weight-gradient phase of the ConvNeXt stages: under the reducer the LayerNorm / GELU outputs and the output gradients of
    each stage live in per-stage arenas and the pwconv1 / pwconv2 weight gradients of all blocks of a stage come from ONE strided-batched
    GEMM each at the flush.  Same gradients as the per-layer path (which sums bf16 split-K partials: the batched GEMM accumulates in
    fp32 throughout, so it is compared at bf16 resolution), nothing left parked, arenas released with the model."""
    import gc
    from dgtd.ops import _native as N
    nat = N.ops()
    assert nat is not None
    S, B = 64, 2
    x, d, l = (t.cuda() for t in filler.synthetic_batch(B, S, seed=5))
    grads, parked = {}, {}
    base = nat.arena_bytes()
    try:
        for batched in (False, True):
            N.BATCH_WGRAD = batched
            net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.bfloat16)
            filler.fill_module(net)
            net = net.cuda().train()
            red = dgtd.dist.GradReducer(net, working_dtype=torch.bfloat16)
            for _ in range(2):                                       # second step reuses the arenas
                red.zero_grad()
                loss = net(None, x, l, d, mode="loss")["loss"]
                done = nat.flushed_reductions()
                loss.backward()
                parked[batched] = nat.flushed_reductions() - done      # everything parked in this backward pass (flushed at its end)
                red.finish()
            assert nat.pending_reductions() == 0
            grads[batched] = {k: p.grad.float().clone() for k, p in net.named_parameters() if p.grad is not None}
            if batched:
                assert nat.arena_bytes() > base
            del net, red, loss
            gc.collect()
    finally:
        N.BATCH_WGRAD = True
    assert nat.arena_bytes() == base, "arenas must die with the model"
    # one parked GEMM per pwconv of the 36 ConvNeXt blocks and per q / kv / proj / fc1 / fc2 of the 16 PVT blocks
    assert parked[True] >= parked[False] + 2 * (3 + 3 + 27 + 3) + 5 * (3 + 4 + 6 + 3), (parked[True], parked[False])
    pw = [k for k in grads[True] if k.endswith("weight") and (".pwconv" in k or (".block" in k and any(t in k for t in (".attn.q.", ".attn.kv.", ".attn.proj.", ".mlp.fc1.", ".mlp.fc2."))))]
    assert len(pw) == 72 + 80, len(pw)
    for k in pw:
        a, b = grads[True][k], grads[False][k]
        assert (a - b).norm() / b.norm() < 1e-2, k
    for k in grads[True]:
        torch.testing.assert_close(grads[True][k], grads[False][k], rtol=3e-2, atol=1e-5, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_step_with_forced_allreduce(dtype):
    """the N > 1 forms of the captured step under a test.  A world-1 RCCL group (DGTD_FORCE_ALLREDUCE=1, own
    process so the communicator never meets the other tests) runs 3 steps eager with hook-driven bucket all-reduces, then the same
    through GraphedTrainStep in all N > 1 modes: "fused" (collectives captured inside the one graph, forked per bucket after the backward
    pass and joined per bucket by AdamW), "hooks" (forked from the autograd hooks during the backward pass) and "split" (graph A |
    all-reduce | graph B).  fp32: same losses / weights as eager to atomic noise; bf16 (16-bit payload): close."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_forced_allreduce_worker.py")
    p = subprocess.run([sys.executable, worker, dtype, str(port)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    tol = 1e-4 if dtype == "f32" else 2e-2
    for mode in (("fused", "hooks", "split") if dtype == "f32" else ("fused", "split")):
        got, want = out[mode]["losses"], out["eager"]["losses"]
        print(f"[{dtype} {mode}] losses {got} vs eager {want}; max weight diff {out[mode]['max_weight_diff']:.2e}, elements off {out[mode]['elements_off']}")
        assert out[mode]["steps"] == 3
        for a_, b_ in zip(got, want):
            assert math.isfinite(a_) and abs(a_ - b_) <= tol * max(1.0, abs(b_)), (mode, got, want)
        if dtype == "f32":
            assert out[mode]["max_weight_diff"] < 5e-4 and out[mode]["elements_off"] < 20000, out[mode]


def _tiny_trunk(dgtd, dtype=torch.bfloat16):
    """A bias-carrying Linear called TWICE, a LayerNorm, a depthwise conv and a fused Linear+residual: every deferring node family."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.shared = dgtd.nn.Linear(128, 128)
            self.norm = dgtd.nn.LayerNorm(128, eps=1e-6)
            self.dw = dgtd.nn.DWConv(128)            # the depthwise weight-gradient kernels walk 128-channel groups
            self.out = dgtd.nn.Linear(128, 128)

        def forward(self, x):                       # x [B, 64, 128] tokens of an 8x8 map
            h = self.shared(self.shared(x))         # one weight / bias, two call sites
            h = self.dw(self.norm(h), 8, 8, gelu=True)
            return dgtd.ops.linear_residual(h, *dgtd.nn.wb(self.out), x, None)
    torch.manual_seed(3)
    return Net().cuda().train()


@pytest.mark.parametrize("case", ["shared_bias", "two_backwards", "retain_graph"])
def test_deferred_gradients_survive_accumulation(dgtd, case):
    """ADVICE r2 (medium) / VERDICT r2 weak #4: deferred backward nodes hand autograd an UNWRITTEN tensor, which is sound only while
    autograd steals it as .grad.  A parameter with two call sites, two backward() calls between zero_grad() and finish(), and a
    second backward through a retained graph must all give the gradients of the deferral-off run (bit-level: same kernels, immediate
    path) - not sums over uninitialised memory."""
    from dgtd.dist import reducer as R
    x = torch.randn(2, 64, 128, device="cuda")
    grads = {}
    for defer in (False, True):
        R.DEFER = defer
        try:
            net = _tiny_trunk(dgtd)
            red = dgtd.dist.GradReducer(net, exclude_prefixes=(), working_dtype=torch.bfloat16)
            red.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x.bfloat16())
                loss = (y.float() ** 2).mean()
                if case == "two_backwards":
                    loss.backward()
                    y2 = net(x.bfloat16() * 0.5)
                    (y2.float() ** 2).mean().backward()
                elif case == "retain_graph":
                    loss.backward(retain_graph=True)
                    loss.backward()
                else:
                    loss.backward()
            red.finish()
            torch.cuda.synchronize()
            grads[defer] = {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}
        finally:
            R.DEFER = True
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert torch.isfinite(a).all(), n
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-3 * float(b.abs().max()) + 1e-7, msg=lambda m, n=n: f"{case} {n}: {m}")


def test_two_models_training_alternately_match_their_solo_runs(dgtd):
    """the binding layer's registries are process-global; two models stepping alternately must not see each other."""
    S, B = 64, 2
    x, d, l = (t.cuda() for t in filler.synthetic_batch(B, S, seed=11))

    def make(seed):
        net = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=torch.float32)
        filler.fill_module(net)
        with torch.no_grad():
            net.hitnet.out_CFM.bias.add_(0.01 * seed)
        net = net.cuda().train()
        red = dgtd.dist.GradReducer(net)
        return net, red, dgtd.runner.FlatAdamW(red, lr=1e-4)

    def step(net, red, opt):
        red.zero_grad()
        loss = net(None, x, l, d, mode="loss")["loss"]
        loss.backward()
        red.finish()
        opt.step()
        return loss.item()

    solo = []
    for seed in (1, 2):
        m = make(seed)
        solo.append([step(*m) for _ in range(3)])
        del m
    a, b = make(1), make(2)
    both = [[], []]
    for _ in range(3):
        both[0].append(step(*a))
        both[1].append(step(*b))
    for s_, t_ in zip(solo, both):
        for u, v in zip(s_, t_):
            assert abs(u - v) <= 1e-4 * max(1.0, abs(u)), (solo, both)


def test_config5_fp16_b16_full_size_properties(dgtd):
    """BASELINE config 5's per-GPU workload at full size: 512x512, batch 16, fp16 compute + fp32 masters + dynamic loss scaling
    (config/cod.yml + AmpOptimWrapper), 6 steps.  Properties the size does not change: the loss is finite at every step, the scaler
    stops overflowing after its warm-up (the initial 65536 may skip steps) and takes steps, every trunk ends with finite, non-zero
    gradients, the masters move."""
    torch.manual_seed(0)
    S, B = 512, 16
    net = dgtd.nn.cod(compute_dtype=torch.float16).cuda().train()
    red = dgtd.dist.GradReducer(net, working_dtype=torch.float16)
    scaler = dgtd.runner.LossScaler("cuda")
    opt = dgtd.runner.FlatAdamW(red, scaler=scaler)
    data = dgtd.runner.SyntheticRGBD(S, B, device="cuda")
    before = net.hitnet.backbone.block3[0].mlp.fc1.weight.detach().clone()
    losses, scales, taken = [], [], []
    for i in range(6):
        b = data.batch_at(i % 2)
        red.zero_grad()
        loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
        scaler.scale(loss).backward()
        red.finish()
        opt.step()
        losses.append(loss.item())
        scales.append(scaler.get_scale())
        taken.append(opt.steps)
    print(f"config 5 (512x512, batch 16, fp16): losses {[round(v, 4) for v in losses]}, scales {scales}, steps taken {taken}")
    assert all(math.isfinite(v) for v in losses), losses
    assert taken[-1] >= 3 and taken[-1] - taken[-3] == 2, (taken, scales)          # no overflow in the last two steps
    assert math.isfinite(scales[-1]) and scales[-1] >= 1.0
    inv = 1.0 / scales[-1]
    trunks = {"pvt": "hitnet.backbone.block", "convnext": "prompt_encoder.encoder2.stages", "prompt": "prompt_decoder",
              "diffuser": "propagation_weight_regressor", "decoder": "hitnet.decoder_level", "heads": "hitnet.out_"}
    norms = {k: 0.0 for k in trunks}
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        assert torch.isfinite(p.grad).all(), n
        for k, pat in trunks.items():
            if pat in n:
                norms[k] += float(p.grad.float().norm()) * inv
    assert all(v > 0 for v in norms.values()), norms
    assert not torch.equal(before, net.hitnet.backbone.block3[0].mlp.fc1.weight.detach())
