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
