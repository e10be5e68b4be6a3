"""Checkpoint plumbing with the reference's file layouts (twig/model/cod.py:237-300, class ``our_init``).

* ``load_pretrained``  = our_init.before_train: ``pretrain/pvt_v2_b2.pth`` -> ``hitnet.backbone`` and
  ``pretrain/convnext_base_22k_224.pth`` -> ``hitnet.backbone.prompt_encoder.encoder2``; either file may be a raw state_dict or
  wrap it under ``['model']`` (cod.py:265-267, :274-277); ``strict=False`` with the report returned instead of printed.
* ``load_checkpoint``  = our_init.before_val: mmengine layout, weights under ``['state_dict']`` (cod.py:299).
* ``save_checkpoint``  writes that layout (``state_dict`` / ``optimizer`` / ``param_schedulers`` / ``meta``), fp32 masters only:
  working copies and DropPath plans are not part of the state_dict.
Parameter names and shapes are identical to the reference's (879 keys), so model weights move in both directions unchanged; the
``optimizer`` entry has torch.optim.AdamW's state_dict layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``) whichever of
the two optimizers wrote it.  A model that already has a ``dist.GradReducer`` keeps computing with working copies: the reducer's
load hooks refresh them whenever a (sub)module's state is loaded."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch


def _unwrap(obj, key):
    return obj[key] if isinstance(obj, dict) and key in obj else obj


def load_pretrained(model: torch.nn.Module, pvt_path: Optional[str] = "pretrain/pvt_v2_b2.pth",
                    convnext_path: Optional[str] = "pretrain/convnext_base_22k_224.pth", map_location="cpu") -> Dict[str, object]:
    report = {}
    if pvt_path is not None:
        ckpt = _unwrap(torch.load(pvt_path, map_location=map_location, weights_only=True), "model")
        report["pvt"] = model.hitnet.backbone.load_state_dict(ckpt, strict=False)
    if convnext_path is not None:
        ckpt = _unwrap(torch.load(convnext_path, map_location=map_location, weights_only=True), "model")
        report["convnext"] = model.hitnet.backbone.prompt_encoder.encoder2.load_state_dict(ckpt, strict=False)
    return report


def load_checkpoint_file(path: str, map_location="cpu", weights_only: bool = True):
    """torch.load restricted to tensors and plain containers by default; ``weights_only=False`` is the explicit opt-out for files
    from a trusted source that pickle other objects (mmengine checkpoints carry such objects under ``meta`` / ``message_hub``)."""
    return torch.load(path, map_location=map_location, weights_only=weights_only)


def load_checkpoint(model: torch.nn.Module, path: str, map_location="cpu", strict: bool = False, weights_only: bool = True):
    ckpt = _unwrap(load_checkpoint_file(path, map_location, weights_only), "model")
    return model.load_state_dict(_unwrap(ckpt, "state_dict"), strict=strict)


def save_checkpoint(model: torch.nn.Module, path: str, optimizer=None, schedulers=None, meta: Optional[dict] = None) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    out = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "meta": dict(meta or {})}
    if optimizer is not None:
        out["optimizer"] = optimizer.state_dict()
    if schedulers is not None:
        out["param_schedulers"] = [s.state_dict() for s in schedulers]
    torch.save(out, path)
