"""CPU: checkpoint layouts (our_init, cod.py:237-300) and the config keys of config/sod.yml / config/cod.yml."""
import math
import os

import pytest
import torch

SOD_LIKE = """
train_cfg:
  by_epoch: &by_epoch True
  max_epochs: &max_epochs 50
  val_interval: 50
find_unused_parameters: True
train_dataloader:
  batch_size: 10
  num_workers: 8
model:
  type: cod
  win_size: 22
  filter_ratio: 0.9
  using_sam: True
  head:
    type: mmseg.models.decode_heads.SegformerHead
    num_classes: 1
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
def model():
    import dgtd
    cfg = dgtd.runner.load_config(SOD_LIKE)
    return cfg, dgtd.runner.build_model(cfg, compute_dtype=torch.float32)


def test_config_builds_model_and_param_groups(model):
    import dgtd
    cfg, net = model
    assert isinstance(net, dgtd.nn.cod)            # model kwargs of the YAML are accepted and ignored (cod.py:38-46)
    opt = dgtd.runner.build_optim(cfg, net)
    lrs = sorted({round(g["lr"], 8) for g in opt.param_groups})
    assert lrs == [0.00001, 0.0001, 0.0005]        # x0.02 ConvNeXt trunk, x0.2 backbone, x1 Hitnet heads
    n_params = sum(len(g["params"]) for g in opt.param_groups)
    assert n_params == len({id(p) for p in net.parameters()})   # bypass_duplicate: the shared PReLU appears once
    by_lr = {round(g["lr"], 8): sum(p.numel() for p in g["params"]) for g in opt.param_groups}
    trunk = sum(p.numel() for n, p in net.named_parameters() if ".encoder2.downsample_layers" in n or ".encoder2.stages." in n)
    assert by_lr[0.00001] == trunk
    sched = dgtd.runner.CosineByEpoch(opt, cfg["param_scheduler"]["T_max"])
    for _ in range(25):
        sched.step()
    assert math.isclose(opt.param_groups[0]["lr"], opt.param_groups[0]["initial_lr"] * 0.5, rel_tol=1e-9)


def test_checkpoint_layouts_roundtrip(model, tmp_path):
    import dgtd
    _, net = model
    # mmengine layout written and read back (our_init.before_val reads ['state_dict'], cod.py:299)
    path = os.path.join(tmp_path, "epoch_1.pth")
    dgtd.runner.save_checkpoint(net, path, meta={"epoch": 1})
    blob = torch.load(path, weights_only=False)
    assert set(blob) >= {"state_dict", "meta"} and len(blob["state_dict"]) == 879
    other = dgtd.nn.cod()
    res = dgtd.runner.load_checkpoint(other, path)
    assert not res.missing_keys and not res.unexpected_keys
    k = "hitnet.backbone.block3.5.attn.kv.weight"
    assert torch.equal(other.state_dict()[k], net.state_dict()[k])
    # pretrain layouts (our_init.before_train): raw dict for PVT, ['model'] wrapper for ConvNeXt, both strict=False
    pvt = {kk[len("hitnet.backbone."):]: v for kk, v in net.state_dict().items()
           if kk.startswith("hitnet.backbone.block") or kk.startswith("hitnet.backbone.patch_embed") or kk.startswith("hitnet.backbone.norm")}
    pvt["head.weight"] = torch.zeros(1000, 512)                       # classifier keys of the ImageNet file are unexpected
    cnx = {kk[len("hitnet.backbone.prompt_encoder.encoder2."):]: v for kk, v in net.state_dict().items()
           if ".encoder2.stages." in kk or ".encoder2.downsample_layers." in kk}
    torch.save(pvt, os.path.join(tmp_path, "pvt.pth"))
    torch.save({"model": cnx}, os.path.join(tmp_path, "cnx.pth"))
    fresh = dgtd.nn.cod()
    rep = dgtd.runner.load_pretrained(fresh, os.path.join(tmp_path, "pvt.pth"), os.path.join(tmp_path, "cnx.pth"))
    assert rep["pvt"].unexpected_keys == ["head.weight"]
    assert all(m.startswith(("prompt_encoder", "prompt_decoder")) for m in rep["pvt"].missing_keys)
    assert all(m.startswith(("convs.", "fusion_conv.")) for m in rep["convnext"].missing_keys)
    kk = "hitnet.backbone.prompt_encoder.encoder2.stages.2.26.pwconv1.weight"
    assert torch.equal(fresh.state_dict()[kk], net.state_dict()[kk])


def test_default_sampler_partitions_a_seeded_permutation_rank_strided():
    """mmengine DefaultSampler (config/sod.yml:24-26): every rank builds the SAME permutation from seed + epoch, pads it by
    repetition to a multiple of the world size and takes indices[rank::world]."""
    import dgtd
    N, world = 10, 4
    per_rank = []
    for r in range(world):
        s = dgtd.runner.DefaultSampler(N, shuffle=True, seed=7, rank=r, world=world)
        s.set_epoch(3)
        per_rank.append(list(s))
        assert len(per_rank[-1]) == len(s) == 3
    g = torch.Generator()
    g.manual_seed(7 + 3)
    perm = torch.randperm(N, generator=g).tolist()
    padded = (perm * 2)[:12]
    assert [padded[r::world] for r in range(world)] == per_rank
    assert sorted(set(sum(per_rank, []))) == list(range(N))           # every sample is seen
    s0 = dgtd.runner.DefaultSampler(N, shuffle=True, seed=7, rank=0, world=world)
    s0.set_epoch(4)
    assert list(s0) != per_rank[0]                                    # the permutation changes with the epoch
    val = dgtd.runner.DefaultSampler(5, shuffle=False, rank=1, world=2)
    assert list(val) == [1, 3, 0]                                     # sequential, round_up wraps to the start
    ds = dgtd.runner.SyntheticRGBD(32, 2, device="cpu", length=5)
    bs = list(dgtd.runner.batches(ds, val, 2, "cpu"))
    assert [len(b["input"]) for b in bs] == [2, 1] and bs[0]["raw"][0].endswith("/1.png")
    assert torch.equal(bs[0]["input"][1], ds[3]["input"])


def test_metric_restatements_match_the_reference_goldens():
    """runner.metrics.mean_iou_reference against values computed by the reference's own meanIntersectionOverUnion.mean_iou
    (twig/metric/mIOU.py:32-58; oracle/make_golden_metrics.py)."""
    import numpy as np
    import dgtd
    from oracle.make_golden_metrics import CASES, GOLDEN, case_inputs
    g = np.load(GOLDEN)
    for case in CASES:
        pred, target = case_inputs(*case)
        got = dgtd.runner.metrics.mean_iou_reference(pred, target)
        assert abs(got - float(g[case[0]])) < 1e-6, case[0]
    assert float(g["c1"]) == 1.0                                      # one output channel: identically 1
    # binary 2-class mIoU: hand-checked 2x2 case: pred>0.5 = [[1,0],[1,1]], gt = [[1,0],[0,1]]
    p = torch.tensor([[[[0.9, 0.1], [0.8, 0.7]]]])
    t = torch.tensor([[[[1.0, 0.0], [0.0, 1.0]]]])
    # class 0: TP 1, FP 0, FN 1 -> 1/2; class 1: TP 2, FP 1, FN 0 -> 2/3
    assert abs(dgtd.runner.metrics.binary_miou(p, t) - (0.5 + 2 / 3) / 2) < 1e-9
    ev = dgtd.runner.metrics.build_evaluators([{"type": "Emeasure"}, {"type": "MAE"}], log=lambda m: None)
    assert [type(e).__name__ for e in ev] == ["MAE"]
    ev[0].process(None, (p, t))
    # uint8 quantisation + min-max normalisation: pred -> [229,25,204,178]/255 -> (x - 25/255)/(204/255)
    q = torch.tensor([229., 25., 204., 178.])
    want = ((q - 25) / 204 - torch.tensor([1., 0., 0., 1.])).abs().mean().item()
    assert abs(ev[0].compute_metrics()["MAE"] - want) < 1e-6


def test_live_reference_miou_when_available():
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("reference tree not present")
    import dgtd
    from oracle.make_golden_metrics import CASES, case_inputs, load_reference_metric
    metric = load_reference_metric()
    for case in CASES:
        pred, target = case_inputs(*case)
        assert abs(float(metric.mean_iou(pred.clone(), target.clone())) - dgtd.runner.metrics.mean_iou_reference(pred, target)) < 1e-6


def test_cosine_schedule_resumes():
    import dgtd
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    a = dgtd.runner.CosineByEpoch(opt, 10)
    for _ in range(4):
        a.step()
    lr4 = opt.param_groups[0]["lr"]
    opt2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    b = dgtd.runner.CosineByEpoch(opt2, 10)
    b.load_state_dict(a.state_dict())
    assert opt2.param_groups[0]["lr"] == lr4
    a.step(); b.step()
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]


def test_export_registry_and_nest_shim():
    """`from nest import export` (twig/dataset/sod_train.py:2,11; twig/metric/MAE.py:2,8; twig/model/cod.py:34-36) resolves
    to the runner's registry; exported classes are built from `type:` entries like mmengine does, and exported metrics are visible
    to val_evaluator."""
    import sys
    import dgtd
    R = dgtd.runner.registry
    saved = sys.modules.pop("nest", None)
    try:
        nest = dgtd.runner.install_nest_shim()
        from nest import export, register_model
        assert nest.export is export is R.export

        @export
        class SOD_TRAIN_LIKE:
            def __init__(self, data_dir, depth_dir, split, image_size=None):
                self.args = (data_dir, depth_dir, split, image_size)

        @export
        class Emeasure:                                   # a metric class the reference exports (twig/metric/Emeasure.py)
            def __init__(self, prefix=None):
                self.prefix = prefix

        @export
        @register_model
        class tiny_model:
            def __init__(self, **kw):
                self.kw = kw

        ds = dgtd.runner.build_exported({"type": "SOD_TRAIN_LIKE", "data_dir": "d", "depth_dir": "dd", "split": "train"})
        assert ds.args == ("d", "dd", "train", None)
        assert R.get("tiny_model") is tiny_model and dgtd.runner.config.MODEL_REGISTRY["tiny_model"] is tiny_model
        assert "SyntheticRGBD" in R.REGISTRY                # the runner's own dataset registers the same way
        logs = []
        evs = dgtd.runner.metrics.build_evaluators([{"type": "MAE"}, {"type": "Emeasure", "prefix": "COD"}, {"type": "Smeasure"}], logs.append)
        assert len(evs) == 2 and isinstance(evs[1], Emeasure) and evs[1].prefix == "COD"      # exported metric wins over "skipped"
        assert any("Smeasure" in m for m in logs)
        with pytest.raises(KeyError, match="nothing exported"):
            R.get("no_such_class")
    finally:
        for k in ("SOD_TRAIN_LIKE", "Emeasure", "tiny_model"):
            R.REGISTRY.pop(k, None)
        dgtd.runner.config.MODEL_REGISTRY.pop("tiny_model", None)
        sys.modules.pop("nest", None)
        if saved is not None:
            sys.modules["nest"] = saved
