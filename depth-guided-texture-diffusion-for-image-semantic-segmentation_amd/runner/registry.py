"""The ``@export`` registry of the reference's (external) ``nest`` package: ``from nest import export`` decorates every class a config
may name by ``type:`` - datasets (twig/dataset/sod_train.py:11 ``SOD_TRAIN``, sod_test.py ``SOD_TEST``), metrics (twig/metric/MAE.py:8,
Emeasure / Fmeasure / Smeasure / mIOU), models (twig/model/cod.py:35 ``@export`` + ``@register_model``), hooks.  ``nest`` itself is not
in the reference tree (requirements.txt:70, a private git URL), so the runner carries the minimal counterpart: classes register by
name, ``build(cfg)`` instantiates ``cfg['type']`` with the remaining keys, and ``install_nest_shim()`` publishes a ``nest`` module with
``export`` / ``register_model`` so that the reference's twig/dataset and twig/metric files import UNCHANGED where their other
dependencies (torchvision, mmengine, py_sod_metrics) exist."""
from __future__ import annotations

import sys
import types
from typing import Any, Callable, Dict

REGISTRY: Dict[str, Callable] = {}


def export(obj=None, *, name: str = None):
    """``@export`` / ``@export(name=...)``: register a class or function under its (or the given) name; returns it unchanged.
    Re-registering the same object is a no-op; a DIFFERENT object under a taken name replaces it (the reference re-imports
    modules freely) - the last definition wins, like a Python module attribute."""
    def deco(o):
        REGISTRY[name or o.__name__] = o
        return o
    return deco if obj is None else deco(obj)


def register_model(obj=None, *, name: str = None):
    """twig/model/cod.py:34-36 stacks ``@register_model`` under ``@export``: same registry, also visible to runner.config.build_model."""
    def deco(o):
        from .config import MODEL_REGISTRY
        MODEL_REGISTRY[name or o.__name__] = o
        return export(o, name=name)
    return deco if obj is None else deco(obj)


def get(name: str) -> Callable:
    try:
        return REGISTRY[name]
    except KeyError:
        raise KeyError(f"nothing exported under {name!r}; known: {sorted(REGISTRY)}") from None


def build(cfg: Dict[str, Any], **extra):
    """mmengine-style: ``{'type': 'SOD_TRAIN', 'data_dir': ...}`` -> ``SOD_TRAIN(data_dir=...)``."""
    cfg = dict(cfg)
    return get(cfg.pop("type"))(**cfg, **extra)


def install_nest_shim() -> types.ModuleType:
    """Make ``from nest import export`` (and ``register_model``) resolve to this registry.  Idempotent; a real ``nest`` wins."""
    mod = sys.modules.get("nest")
    if mod is None:
        mod = types.ModuleType("nest")
        mod.__doc__ = "dgtd.runner.registry shim of the reference's external `nest` package (export / register_model only)"
        mod.export, mod.register_model = export, register_model
        sys.modules["nest"] = mod
    return mod
