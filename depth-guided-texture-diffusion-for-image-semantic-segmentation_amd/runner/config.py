"""Minimal config-driven runner for the keys the reference's YAML files use (config/sod.yml:1-104, config/cod.yml:1-143):
``train_cfg`` (by_epoch, max_epochs, val_interval), ``train_dataloader.batch_size``, ``model.type`` (+ kwargs the reference
ignores, cod.py:38-46), ``optim_wrapper`` (AdamW, ``paramwise_cfg.custom_keys`` lr_mult, bypass_duplicate),
``param_scheduler`` (CosineAnnealingLR by epoch), ``default_hooks.logger.interval`` / ``checkpoint.interval``,
``custom_hooks: our_init``, ``*_dataloader.sampler`` (DefaultSampler), ``val_evaluator``, ``optim_wrapper.type: AmpOptimWrapper``
(= fp16 autocast + dynamic loss scaling when the runner is built with ``compute_dtype=torch.float16``).  The external nest/mmengine runner is out of scope; this is what drives the HIP-backed model from
the same file.  Datasets need real data, so the loop is fed by any iterable of the dataset dict contract (runner/data.py)."""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, Optional

import torch
import yaml

from .checkpoint import load_checkpoint_file, load_pretrained, save_checkpoint
from .data import DefaultSampler, batches
from .metrics import build_evaluators
from .optim import FlatAdamW, LossScaler, build_optimizer

MODEL_REGISTRY: Dict[str, Callable] = {}


def register_model(name: str):
    def deco(fn):
        MODEL_REGISTRY[name] = fn
        return fn
    return deco


def load_config(path_or_text: str) -> dict:
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    return yaml.safe_load(text)


def build_model(cfg: dict, compute_dtype=torch.bfloat16):
    from ..nn import cod
    MODEL_REGISTRY.setdefault("cod", cod)
    mcfg = dict(cfg["model"])
    cls = MODEL_REGISTRY[mcfg.pop("type")]
    return cls(**mcfg, compute_dtype=compute_dtype)


def build_optim(cfg: dict, model: torch.nn.Module, capturable: bool = False):
    ow = cfg["optim_wrapper"]
    ocfg = dict(ow["optimizer"])
    assert ocfg.pop("type") == "AdamW", "the reference configs use AdamW (config/sod.yml:58-61)"
    keys = {k: float(v.get("lr_mult", 1.0)) for k, v in (ow.get("paramwise_cfg", {}).get("custom_keys") or {}).items()}
    return build_optimizer(model, lr=float(ocfg["lr"]), weight_decay=float(ocfg.get("weight_decay", 0.0)), custom_keys=keys,
                           fused=None, capturable=capturable)


class CosineByEpoch:
    """CosineAnnealingLR(by_epoch=True, T_max=max_epochs, eta_min=0) (config/sod.yml:78-81): lr_e = lr_0 * (1 + cos(pi e / T)) / 2."""

    def __init__(self, optimizer, t_max: int):
        self.opt, self.t_max, self.epoch = optimizer, int(t_max), 0
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])

    def _apply(self):
        f = 0.5 * (1.0 + math.cos(math.pi * min(self.epoch, self.t_max) / self.t_max))
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * f

    def step(self):
        self.epoch += 1
        self._apply()

    def state_dict(self):
        return {"epoch": self.epoch, "t_max": self.t_max}

    def load_state_dict(self, sd):
        self.epoch, self.t_max = int(sd["epoch"]), int(sd["t_max"])
        self._apply()


def custom_keys_of(cfg: dict) -> Dict[str, float]:
    ow = cfg["optim_wrapper"]
    return {k: float(v.get("lr_mult", 1.0)) for k, v in (ow.get("paramwise_cfg", {}).get("custom_keys") or {}).items()}


class Runner:
    """Train / validate the HIP-backed model from the reference's YAML.  On the HIP device the optimizer is ``FlatAdamW`` over the
    gradient reducer's flat buckets (what bench.py measures); ``compute_dtype=torch.float16`` adds the dynamic ``LossScaler`` of
    the configs' ``AmpOptimWrapper``.  On CPU (gloo tests of the host logic) it is torch.optim.AdamW over the same groups."""

    def __init__(self, cfg: dict, device="cuda", compute_dtype=torch.bfloat16, work_dir="work_dir", log: Callable = print,
                 rank: int = 0, world: int = 1, seed: int = 0):
        from ..dist import GradReducer, broadcast_parameters
        self.cfg, self.device, self.work_dir, self.log = cfg, device, work_dir, log
        self.rank, self.world, self.seed = rank, world, seed
        self.model = build_model(cfg, compute_dtype).to(device)
        if any(h.get("type") == "our_init" for h in cfg.get("custom_hooks") or []):
            have = [p for p in ("pretrain/pvt_v2_b2.pth", "pretrain/convnext_base_22k_224.pth") if os.path.exists(p)]
            if len(have) == 2:
                self.log(load_pretrained(self.model))
            else:
                self.log("our_init: pretrain/*.pth not found, keeping the random initialisation")
        broadcast_parameters(self.model)
        self.reducer = GradReducer(self.model, working_dtype=compute_dtype)
        ow = cfg["optim_wrapper"]
        ocfg = dict(ow["optimizer"])
        assert ocfg.get("type") == "AdamW", "the reference configs use AdamW (config/sod.yml:58-61)"
        on_gpu = torch.device(device).type == "cuda"
        self.scaler = None
        if on_gpu:
            if compute_dtype == torch.float16:
                assert ow.get("type") == "AmpOptimWrapper", "fp16 compute needs the AmpOptimWrapper recipe (loss scaling)"
                self.scaler = LossScaler(device)
            self.optimizer = FlatAdamW(self.reducer, lr=float(ocfg["lr"]), weight_decay=float(ocfg.get("weight_decay", 0.0)),
                                       custom_keys=custom_keys_of(cfg), scaler=self.scaler)
        else:
            self.optimizer = build_optim(cfg, self.model)
        tc = cfg["train_cfg"]
        self.max_epochs = int(tc["max_epochs"])
        self.val_interval = int(tc.get("val_interval", 0) or 0)
        ps = cfg.get("param_scheduler") or {}
        self.scheduler = CosineByEpoch(self.optimizer, int(ps.get("T_max", self.max_epochs))) if ps.get("type") == "CosineAnnealingLR" else None
        hooks = cfg.get("default_hooks") or {}
        self.log_interval = int((hooks.get("logger") or {}).get("interval", 50))
        self.ckpt_interval = int((hooks.get("checkpoint") or {}).get("interval", 1))
        self.evaluators = build_evaluators(cfg.get("val_evaluator"), log)
        self.epoch = 0

    # ------------------------------------------------------------------ data
    def loader(self, dataset, split: str = "train", epoch: int = 0):
        """Batches of ``dataset`` the way the YAML's ``<split>_dataloader`` asks: batch_size, DefaultSampler(shuffle) partitioned
        ``rank::world`` over a per-epoch seeded permutation."""
        dl = self.cfg.get(f"{split}_dataloader") or {}
        sc = dl.get("sampler") or {}
        sampler = DefaultSampler(len(dataset), shuffle=bool(sc.get("shuffle", split == "train")), seed=self.seed, rank=self.rank, world=self.world)
        sampler.set_epoch(epoch)
        return batches(dataset, sampler, int(dl.get("batch_size", 1)), self.device)

    # ------------------------------------------------------------------ train
    def train_step(self, batch: dict) -> torch.Tensor:
        self.reducer.zero_grad()
        loss = self.model(batch.get("raw"), batch["input"], batch["label"], batch["depth"], mode="loss")["loss"]
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        self.reducer.finish()
        self.optimizer.step()
        if not isinstance(self.optimizer, FlatAdamW):
            self.reducer.refresh_working()
        return loss

    def train(self, loader_fn: Callable[[int], Iterable[dict]], epochs: Optional[int] = None, val_loader_fn: Optional[Callable[[], Iterable[dict]]] = None):
        """Epochs ``self.epoch .. (epochs or max_epochs)``: the loop the reference gets from mmengine's EpochBasedTrainLoop
        (logger / checkpoint hooks by their intervals, scheduler per epoch, validation every ``val_interval`` epochs)."""
        losses = []
        for epoch in range(self.epoch, epochs or self.max_epochs):
            self.model.train()
            for it, batch in enumerate(loader_fn(epoch)):
                loss = self.train_step(batch)
                losses.append(loss.detach())
                if (it + 1) % self.log_interval == 0:
                    self.log(f"epoch {epoch + 1} iter {it + 1} loss {loss.item():.4f} lr {self.optimizer.param_groups[0]['lr']:.3e}")
            if self.scheduler is not None:
                self.scheduler.step()
            self.epoch = epoch + 1
            if self.epoch % self.ckpt_interval == 0 and self.rank == 0:
                self.save(os.path.join(self.work_dir, f"epoch_{self.epoch}.pth"))
            if val_loader_fn is not None and self.val_interval and self.epoch % self.val_interval == 0:
                self.log(f"epoch {self.epoch} val {self.validate(val_loader_fn())}")
        return [float(l) for l in losses]

    # ------------------------------------------------------------------ val (script/test.sh: `-m val`)
    @torch.no_grad()
    def validate(self, loader: Iterable[dict]) -> Dict[str, float]:
        """val_step = ``cod.forward(mode='predict')`` (cod.py:152-153, :219) per batch -> every evaluator's ``process`` ->
        ``compute_metrics`` (mmengine ValLoop + Evaluator)."""
        was_training = self.model.training
        self.model.eval()
        for ev in self.evaluators:
            ev.results.clear()
            ev.__dict__.pop("_all", None)
        for batch in loader:
            out = self.model(batch.get("raw"), batch["input"], batch["label"], batch["depth"], mode="predict")
            for ev in self.evaluators:
                ev.process(batch, out)
        self.model.train(was_training)
        metrics = {}
        for ev in self.evaluators:
            metrics.update(ev.compute_metrics())
        return metrics

    # ------------------------------------------------------------------ checkpoint / resume
    def save(self, path: str) -> None:
        save_checkpoint(self.model, path, self.optimizer, [self.scheduler] if self.scheduler else None, {"epoch": self.epoch})

    def resume(self, path: str) -> None:
        """Continue a run: weights (the reducer's load hooks refresh the working copies), optimizer state (moments, step, loss
        scale), scheduler and epoch."""
        ckpt = load_checkpoint_file(path, weights_only=False)
        self.model.load_state_dict(ckpt["state_dict"], strict=True)
        if "optimizer" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        if self.scheduler is not None and ckpt.get("param_schedulers"):
            self.scheduler.load_state_dict(ckpt["param_schedulers"][0])
        self.epoch = int((ckpt.get("meta") or {}).get("epoch", 0))
