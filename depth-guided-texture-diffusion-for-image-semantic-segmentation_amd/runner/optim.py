"""AdamW with the reference's per-prefix learning-rate multipliers (config/sod.yml:56-76, config/cod.yml):
mmengine's ``paramwise_cfg.custom_keys`` semantics = the LONGEST matching substring key wins;
``bypass_duplicate`` = a parameter reachable under several names (the shared PReLU) is added once."""
from __future__ import annotations

from typing import Dict, Optional

import torch

SOD_CUSTOM_KEYS: Dict[str, float] = {
    "hitnet.backbone": 0.2,
    "hitnet.backbone.prompt_encoder.encoder2.downsample_layers": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.0": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.1": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.2": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.3": 0.02,
}


def lr_mult_for(name: str, custom_keys: Dict[str, float]) -> float:
    best, mult = -1, 1.0
    for key, m in custom_keys.items():
        if key in name and len(key) > best:
            best, mult = len(key), m
    return mult


def build_optimizer(model: torch.nn.Module, lr: float = 5e-4, weight_decay: float = 0.1,
                    custom_keys: Optional[Dict[str, float]] = None, fused: Optional[bool] = None,
                    capturable: bool = False):
    custom_keys = SOD_CUSTOM_KEYS if custom_keys is None else custom_keys
    groups: Dict[float, list] = {}
    seen = set()
    for name, p in model.named_parameters(remove_duplicate=False):
        if id(p) in seen:   # bypass_duplicate; masters whose compute runs on a working copy have requires_grad=False
            continue
        seen.add(id(p))
        groups.setdefault(lr_mult_for(name, custom_keys), []).append(p)
    param_groups = [{"params": ps, "lr": lr * m, "initial_lr": lr * m} for m, ps in sorted(groups.items(), reverse=True)]
    if fused is None:
        fused = any(p.is_cuda for g in param_groups for p in g["params"])
    return torch.optim.AdamW(param_groups, lr=lr, weight_decay=weight_decay, fused=fused, capturable=capturable and fused)
