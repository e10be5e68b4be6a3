"""Synthetic RGB-D batches with the reference datasets' dict contract (twig/dataset/sod_train.py:55-83):
{'raw': str, 'input': [3,S,S] ImageNet-normalised, 'label': [1,S,S] in {0,1}, 'depth': [1,S,S] in [0,1]},
delivered as LISTS of per-sample tensors the way mmengine's pseudo_collate hands them to cod.forward."""
from __future__ import annotations

import torch
import torch.nn.functional as F


class SyntheticRGBD:
    def __init__(self, size: int, batch: int, rank: int = 0, device="cuda", seed: int = 1234):
        self.size, self.batch, self.rank, self.device, self.seed = size, batch, rank, device, seed

    def sample(self, index: int):
        g = torch.Generator(device="cpu").manual_seed(self.seed + self.rank * 10 ** 6 + index)
        S, lo = self.size, max(self.size // 16, 2)
        x = torch.randn(3, S, S, generator=g)
        d = F.interpolate(torch.rand(1, 1, lo, lo, generator=g), size=(S, S), mode="bilinear", align_corners=False)[0]
        l = (F.interpolate(torch.rand(1, 1, lo, lo, generator=g), size=(S, S), mode="bilinear", align_corners=False)[0] > 0.6).float()
        return {"raw": f"synthetic/{self.rank}/{index}.png", "input": x, "label": l, "depth": d.clamp_(0, 1)}

    def batch_at(self, step: int):
        items = [self.sample(step * self.batch + i) for i in range(self.batch)]
        to = lambda k: [it[k].to(self.device, non_blocking=True) for it in items]
        return {"raw": [it["raw"] for it in items], "input": to("input"), "label": to("label"), "depth": to("depth")}
