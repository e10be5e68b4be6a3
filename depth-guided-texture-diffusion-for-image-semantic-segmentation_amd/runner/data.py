"""Synthetic RGB-D batches with the reference datasets' dict contract (twig/dataset/sod_train.py:55-83):
{'raw': str, 'input': [3,S,S] ImageNet-normalised, 'label': [1,S,S] in {0,1}, 'depth': [1,S,S] in [0,1]},
delivered as LISTS of per-sample tensors the way mmengine's pseudo_collate hands them to cod.forward."""
from __future__ import annotations

import torch
import torch.nn.functional as F


class DefaultSampler:
    """mmengine.dataset.DefaultSampler as both configs select it (config/sod.yml:24-26, :35-37): a permutation seeded with
    ``seed + epoch`` (identical on every rank) when ``shuffle``, padded by repetition to a multiple of the world size
    (``round_up``), of which rank r takes ``indices[r::world]``."""

    def __init__(self, length: int, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1, round_up: bool = True):
        self.length, self.shuffle, self.seed, self.rank, self.world, self.round_up = int(length), shuffle, int(seed), rank, world, round_up
        self.epoch = 0
        if round_up:
            self.num_samples = -(-self.length // world)
            self.total_size = self.num_samples * world
        else:
            self.num_samples = -(-(self.length - rank) // world)
            self.total_size = self.length

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(self.length, generator=g).tolist()
        else:
            indices = list(range(self.length))
        if self.round_up:
            indices = (indices * int(self.total_size / len(indices) + 1))[:self.total_size]
        return iter(indices[self.rank:self.total_size:self.world])


def batches(dataset, sampler, batch_size: int, device=None, drop_last: bool = False):
    """Batches in the form mmengine's pseudo_collate hands to ``cod.forward``: a dict of per-key LISTS of per-sample values."""
    cur = []
    for idx in sampler:
        cur.append(dataset[idx])
        if len(cur) == batch_size:
            yield _collate(cur, device)
            cur = []
    if cur and not drop_last:
        yield _collate(cur, device)


def _collate(items, device):
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        out[k] = [v.to(device, non_blocking=True) for v in vals] if (device is not None and torch.is_tensor(vals[0])) else vals
    return out


from .registry import export  # noqa: E402


@export
class SyntheticRGBD:
    def __init__(self, size: int, batch: int, rank: int = 0, device="cuda", seed: int = 1234, length: int = 1 << 20):
        self.size, self.batch, self.rank, self.device, self.seed, self.length = size, batch, rank, device, seed, length

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, index: int):
        """Dataset view (twig/dataset/sod_train.py:55-83 contract): the sample depends on the index only; which rank sees it is
        the sampler's business."""
        rank, self.rank = self.rank, 0
        try:
            return self.sample(index)
        finally:
            self.rank = rank

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


# ------------------------------------------------------------------------------------------------ device-side input pipeline
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def device_preprocess(img_u8: torch.Tensor, size: int, normalize: bool = False, flip: bool = False,
                      out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """One decoded image, uint8 [H, W, C] or [H, W] on the HIP device -> float [C, size, size]: the reference's per-sample transform
    chain RandomHorizontalFlip -> Resize((size, size)) (Pillow BILINEAR, antialiased) -> ToTensor (-> Normalize with the ImageNet
    statistics) of twig/dataset/sod_train.py:31-54, bit-exact, as three small launches (csrc/preprocess.hip)."""
    import ctypes
    from .. import _lib as L
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda:
        raise L.DgtdError("device_preprocess takes a uint8 image on the HIP device")
    x = img_u8.contiguous()
    if x.ndim == 2:
        x = x.unsqueeze(-1)
    H, W, Cc = x.shape
    out = torch.empty(Cc, size, size, dtype=out_dtype, device=x.device)
    ws = torch.empty(L.load().dgtd_preprocess_workspace(H, W, Cc, size), dtype=torch.uint8, device=x.device)
    if normalize:
        if Cc != 3:
            raise L.DgtdError("Normalize applies to the 3-channel RGB image (sod_train.py:35-36)")
        mean = (ctypes.c_float * 3)(*IMAGENET_MEAN)
        std = (ctypes.c_float * 3)(*IMAGENET_STD)
    else:
        mean = std = None
    L.call("dgtd_preprocess", L.ptr(x), L.ptr(out), mean, std, L.ptr(ws), H, W, Cc, size, int(flip), L.dtype_code(out), L.stream_ptr())
    return out


def device_sample(rgb_u8: torch.Tensor, gt_u8: torch.Tensor, depth_u8: torch.Tensor, size: int, flip: bool,
                  out_dtype: torch.dtype = torch.float32) -> dict:
    """The dict of sod_train.py:76-83 for one sample; ``flip`` is the ONE coin the reference shares between image, GT and depth
    by re-seeding before each transform (:65-75)."""
    return {"input": device_preprocess(rgb_u8, size, True, flip, out_dtype), "label": device_preprocess(gt_u8, size, False, flip, out_dtype),
            "depth": device_preprocess(depth_u8, size, False, flip, out_dtype)}
