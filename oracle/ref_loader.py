"""TEST INFRASTRUCTURE — build-container only.  Imports the reference's own
``twig/model/cod.py`` from ``/root/reference`` through inert stub modules so that
golden vectors can be generated from the *real* reference.

Nothing in here ships to or runs on the GPU box: ``/root/reference`` does not exist
there.  Only ``oracle/make_golden.py`` and ``tests/test_oracle_vs_reference.py``
(skipped when the reference tree is absent) use it.

The stubs carry no reference source: they are empty ``types.ModuleType`` objects plus the
three helpers the reference really calls from timm 0.6.13 (``DropPath``, ``to_2tuple``,
``trunc_normal_``; reference call site twig/model/cod.py:816).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("DGTD_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "twig", "model", "cod.py"))


class _DropPath(nn.Module):
    """timm 0.6.13 semantics: identity when p==0 or eval; per-sample Bernoulli otherwise."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _identity(x=None, *a, **k):
    return x


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package so dotted sub-imports resolve
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _install_stubs() -> None:
    class _Empty:  # placeholder base classes
        pass

    stubs = {
        "timm": _stub("timm", create_model=None),
        "timm.models": _stub("timm.models"),
        "timm.models.resnet": _stub("timm.models.resnet", Bottleneck=object),
        "timm.models.layers": _stub("timm.models.layers", DropPath=_DropPath, to_2tuple=_to_2tuple,
                                    trunc_normal_=torch.nn.init.trunc_normal_),
        "timm.models.registry": _stub("timm.models.registry", register_model=_identity),
        "timm.models.vision_transformer": _stub("timm.models.vision_transformer", _cfg=_identity),
        "mmengine": _stub("mmengine"),
        "mmengine.model": _stub("mmengine.model", BaseModel=nn.Module, MMDistributedDataParallel=_Empty),
        "mmengine.hooks": _stub("mmengine.hooks", Hook=object),
        "nest": _stub("nest", export=_identity),
        "transformers": _stub("transformers", AutoImageProcessor=None, DPTForDepthEstimation=None),
        "segment_anything": _stub("segment_anything", sam_model_registry={}),
        "segment_anything.utils": _stub("segment_anything.utils"),
        "segment_anything.utils.transforms": _stub("segment_anything.utils.transforms", ResizeLongestSide=None),
        "torchcam": _stub("torchcam"),
        "torchcam.methods": _stub("torchcam.methods", CAM=None),
        "cv2": _stub("cv2"),
        "mmseg": _stub("mmseg"),
        "torchvision": _stub("torchvision"),
        "torchvision.transforms": _stub("torchvision.transforms"),
        "torchvision.utils": _stub("torchvision.utils", save_image=None),
    }
    for name, mod in stubs.items():
        if name not in sys.modules or name == "transformers":
            sys.modules[name] = mod


_REF = None


def load_reference_cod():
    """Returns the reference's ``cod`` module object (its classes: cod, Hitnet, Attention, …)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise FileNotFoundError(f"reference tree not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    # The one hard-coded device pin on the loss path (twig/model/cod.py:1259).
    torch.Tensor.cuda = lambda self, *a, **k: self
    path = os.path.join(REFERENCE_ROOT, "twig", "model", "cod.py")
    spec = importlib.util.spec_from_file_location("ref_cod", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _REF = mod
    return mod


def build_reference_model(img_size: int, train: bool = False):
    """cod() from the reference with the message-passing output size set to ``img_size``
    (twig/model/cod.py:1252 pins 384) and every DropPath disabled in train mode."""
    ref = load_reference_cod()
    net = ref.cod()
    net.hitnet.backbone.prompt_encoder.message_passing.img_size = img_size
    if train:
        net.train()
        for m in net.modules():
            if isinstance(m, _DropPath):
                m.drop_prob = 0.0
    else:
        net.eval()
    return net
