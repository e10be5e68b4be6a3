"""Minimal config-driven runner for the keys the reference's YAML files use (config/sod.yml:1-104, config/cod.yml:1-143):
``train_cfg`` (by_epoch, max_epochs, val_interval), ``train_dataloader.batch_size``, ``model.type`` (+ kwargs the reference
ignores, cod.py:38-46), ``optim_wrapper`` (AdamW, ``paramwise_cfg.custom_keys`` lr_mult, bypass_duplicate),
``param_scheduler`` (CosineAnnealingLR by epoch), ``default_hooks.logger.interval`` / ``checkpoint.interval``,
``custom_hooks: our_init``.  The external nest/mmengine runner is out of scope; this is what drives the HIP-backed model from
the same file.  Datasets need real data, so the loop is fed by any iterable of the dataset dict contract (runner/data.py)."""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, Optional

import torch
import yaml

from .checkpoint import load_pretrained, save_checkpoint
from .optim import build_optimizer

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

    def step(self):
        self.epoch += 1
        f = 0.5 * (1.0 + math.cos(math.pi * min(self.epoch, self.t_max) / self.t_max))
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * f

    def state_dict(self):
        return {"epoch": self.epoch, "t_max": self.t_max}


class Runner:
    def __init__(self, cfg: dict, device="cuda", compute_dtype=torch.bfloat16, work_dir="work_dir", log: Callable = print):
        from ..dist import GradReducer, broadcast_parameters
        self.cfg, self.device, self.work_dir, self.log = cfg, device, work_dir, log
        self.model = build_model(cfg, compute_dtype).to(device)
        if any(h.get("type") == "our_init" for h in cfg.get("custom_hooks") or []):
            have = [p for p in ("pretrain/pvt_v2_b2.pth", "pretrain/convnext_base_22k_224.pth") if os.path.exists(p)]
            if len(have) == 2:
                self.log(load_pretrained(self.model))
            else:
                self.log("our_init: pretrain/*.pth not found, keeping the random initialisation")
        broadcast_parameters(self.model)
        self.reducer = GradReducer(self.model, working_dtype=compute_dtype)
        self.optimizer = build_optim(cfg, self.model)
        tc = cfg["train_cfg"]
        self.max_epochs = int(tc["max_epochs"])
        ps = cfg.get("param_scheduler") or {}
        self.scheduler = CosineByEpoch(self.optimizer, int(ps.get("T_max", self.max_epochs))) if ps.get("type") == "CosineAnnealingLR" else None
        hooks = cfg.get("default_hooks") or {}
        self.log_interval = int((hooks.get("logger") or {}).get("interval", 50))
        self.ckpt_interval = int((hooks.get("checkpoint") or {}).get("interval", 1))

    def train_step(self, batch: dict) -> torch.Tensor:
        self.reducer.zero_grad()
        loss = self.model(batch.get("raw"), batch["input"], batch["label"], batch["depth"], mode="loss")["loss"]
        loss.backward()
        self.reducer.finish()
        self.optimizer.step()
        self.reducer.refresh_working()
        return loss

    def train(self, loader_fn: Callable[[int], Iterable[dict]], epochs: Optional[int] = None):
        self.model.train()
        for epoch in range(epochs or self.max_epochs):
            for it, batch in enumerate(loader_fn(epoch)):
                loss = self.train_step(batch)
                if (it + 1) % self.log_interval == 0:
                    self.log(f"epoch {epoch + 1} iter {it + 1} loss {loss.item():.4f} lr {self.optimizer.param_groups[0]['lr']:.3e}")
            if self.scheduler is not None:
                self.scheduler.step()
            if (epoch + 1) % self.ckpt_interval == 0:
                save_checkpoint(self.model, os.path.join(self.work_dir, f"epoch_{epoch + 1}.pth"), self.optimizer,
                                [self.scheduler] if self.scheduler else None, {"epoch": epoch + 1})
