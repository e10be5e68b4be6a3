"""TEST INFRASTRUCTURE - build container only.  Golden vectors for the metric restatement (runner/metrics.py): the reference's
own ``meanIntersectionOverUnion.mean_iou`` (twig/metric/mIOU.py:32-58) is imported from /root/reference (inert stubs for ``nest``
and ``mmengine.evaluator``) and run on seeded inputs; inputs are regenerated from the seeds at test time, only the expected
values are stored in tests/golden/metrics.npz.

    python -m oracle.make_golden_metrics
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

from oracle.ref_loader import REFERENCE_ROOT

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "metrics.npz")
# (name, N, C, H, W, seed): C = 1 is the model's case (identically 1.0); C > 1 exercises the confusion-matrix arithmetic
CASES = [("c1", 2, 1, 16, 16, 1), ("c2", 2, 2, 12, 20, 2), ("c3", 1, 3, 24, 24, 3), ("c2_const", 1, 2, 8, 8, 4)]


def case_inputs(name, N, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(N, C, H, W, generator=g)
    # targets are k/255 so that target*255 lands on class ids (mIOU.py:44-46), plus values above C-1 that get clamped
    target = torch.randint(0, C + 2, (N, 1, H, W), generator=g).float() / 255
    if name.endswith("const"):
        target.zero_()
    return pred, target


def load_reference_metric():
    class _Base:
        def __init__(self, collect_device="cpu", prefix=None):
            self.collect_device, self.prefix, self.results = collect_device, prefix, []
    sys.modules.setdefault("nest", types.ModuleType("nest")).export = lambda x=None, *a, **k: x
    ev = types.ModuleType("mmengine.evaluator")
    ev.BaseMetric = _Base
    sys.modules.setdefault("mmengine", types.ModuleType("mmengine"))
    sys.modules["mmengine.evaluator"] = ev
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_miou", os.path.join(REFERENCE_ROOT, "twig", "metric", "mIOU.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.meanIntersectionOverUnion()


def main():
    metric = load_reference_metric()
    out = {}
    for case in CASES:
        pred, target = case_inputs(*case)
        out[case[0]] = np.float64(float(metric.mean_iou(pred.clone(), target.clone())))
        print(case[0], out[case[0]])
    np.savez(GOLDEN, **out)


if __name__ == "__main__":
    main()
