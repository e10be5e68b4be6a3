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
