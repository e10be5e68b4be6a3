"""Evaluators for the validation loop, with mmengine's BaseMetric contract as the reference uses it: ``process(data_batch,
(pred, gt))`` per batch with ``pred = sigmoid(logit)`` and ``gt`` as returned by ``cod.forward(mode='predict')``
(twig/model/cod.py:219), then ``compute_metrics()``.

* ``MeanIoU``  = twig/metric/mIOU.py:32-58 restated on the device (confusion matrix by ``bincount`` instead of the reference's
  per-pixel Python loop).  With the model's single output channel it is identically 1.0 (softmax over one channel -> argmax 0,
  target clamped to class 0) - kept so that the sanity value can be reported for both sides.
* ``BinaryMIoU`` = the 2-class mIoU of ``pred > 0.5`` against ``gt > 0.5`` (SURVEY §8(d) "mIoU definition" (ii)): the figure the
  parity bar "mIoU within 0.1" is quoted on.
* ``MAE``      = twig/metric/MAE.py:18-33: both maps quantised to uint8, then py_sod_metrics.MAE (third-party, pinned at 1.3.1 in
  requirements.txt:110, absent from the reference tree): gt > 128, pred / 255 min-max normalised per image, mean |pred - gt|.
  Restated from the package's published algorithm; PARITY UNPINNED (no reference fixture exists for it).
* E/F/S-measure (config/sod.yml:85-88) are pure py_sod_metrics arithmetic and are not restated: the runner logs that they are
  skipped."""
from __future__ import annotations

from typing import Dict, List

import torch


class _Metric:
    name = "metric"

    def __init__(self):
        self.results: List[float] = []

    def compute_metrics(self) -> Dict[str, float]:
        n = max(1, len(self.results))
        return {self.name: float(sum(self.results) / n)}


def confusion_matrix(pred_idx: torch.Tensor, target_idx: torch.Tensor, num_classes: int) -> torch.Tensor:
    """results[i, j] = number of pixels with target i predicted as j (mIOU.py:17-30)."""
    assert int(pred_idx.max()) < num_classes and int(target_idx.max()) < num_classes
    flat = target_idx.flatten().long() * num_classes + pred_idx.flatten().long()
    return torch.bincount(flat, minlength=num_classes * num_classes).view(num_classes, num_classes)


def mean_iou_reference(pred: torch.Tensor, target: torch.Tensor) -> float:
    """mIOU.py:32-58: pred [N,C,H,W] float, target [N,1,H,W] in [0,1]."""
    assert pred.ndim == 4 and target.ndim == 4
    num_classes = pred.shape[1]
    arg_max = torch.argmax(torch.softmax(pred.float(), dim=1), dim=1)
    t = target.squeeze(1).float() * 255
    t = torch.clamp(t, max=num_classes - 1).long()          # target[target > C-1] = C-1, then truncation to long
    cm = confusion_matrix(arg_max, t, num_classes).double()
    result = 0.0
    for i in range(num_classes):
        nii = cm[i, i]
        if nii == 0:
            continue
        result += float(nii / (cm[i, :].sum() + cm[:, i].sum() - nii))
    return result / num_classes


def binary_miou(prob: torch.Tensor, label: torch.Tensor) -> float:
    """2-class mIoU of (prob > 0.5) vs (label > 0.5), classes absent from both maps skipped like mIOU.py:50-52."""
    cm = confusion_matrix((prob > 0.5).long(), (label > 0.5).long(), 2).double()
    total, present = 0.0, 0
    for i in range(2):
        nii = cm[i, i]
        denom = cm[i, :].sum() + cm[:, i].sum() - nii
        if denom == 0:
            continue
        present += 1
        total += float(nii / denom)
    return total / max(1, present)


class MeanIoU(_Metric):
    name = "mIOU"

    def process(self, data_batch, data_samples) -> None:
        pred, gt = data_samples
        self.results.append(mean_iou_reference(pred, gt))


class BinaryMIoU(_Metric):
    name = "mIoU2"

    def process(self, data_batch, data_samples) -> None:
        pred, gt = data_samples
        self.results.append(binary_miou(pred, gt))


class MAE(_Metric):
    name = "MAE"

    def process(self, data_batch, data_samples) -> None:
        pred, gt = data_samples
        p8 = (pred.squeeze(1).float() * 255).to(torch.uint8)          # MAE.py:25-26 (astype(uint8) truncates)
        g8 = (gt.squeeze(1).float() * 255).to(torch.uint8)
        per_image = []
        for x, y in zip(p8, g8):
            g = (y > 128).float()
            p = x.float() / 255
            lo, hi = p.min(), p.max()
            if hi != lo:
                p = (p - lo) / (hi - lo)
            per_image.append(float((p - g).abs().mean()))
        self._all = getattr(self, "_all", []) + per_image
        self.results.append(sum(self._all) / len(self._all))     # evaluator.get_results() is the running mean over every image so far


EVALUATORS = {"MAE": MAE, "meanIntersectionOverUnion": MeanIoU, "BinaryMIoU": BinaryMIoU}
THIRD_PARTY = ("Emeasure", "Fmeasure", "Smeasure", "WeightedFmeasure")   # pure py_sod_metrics arithmetic: skipped with a message


def build_evaluators(cfg_list, log=print):
    """``val_evaluator`` entries by ``type``: this module's restated metrics first, then anything ``@export``-ed under that name
    (runner/registry.py: the reference's twig/metric classes register themselves there when they can be imported)."""
    from . import registry
    out = []
    for item in cfg_list or []:
        t = item.get("type") if isinstance(item, dict) else str(item)
        if t in EVALUATORS:
            out.append(EVALUATORS[t]())
        elif t in registry.REGISTRY:
            out.append(registry.build(item if isinstance(item, dict) else {"type": t}))
        elif t in THIRD_PARTY:
            log(f"val_evaluator {t}: third-party py_sod_metrics arithmetic, not restated here - skipped")
        else:
            raise KeyError(f"unknown val_evaluator type {t!r}")
    return out
