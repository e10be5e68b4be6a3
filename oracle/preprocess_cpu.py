"""TEST INFRASTRUCTURE — CPU restatement of the reference's per-sample input pipeline
(twig/dataset/sod_train.py:31-54 transforms, :65-83 application): RandomHorizontalFlip -> Resize((S, S)) -> ToTensor
(-> Normalize(ImageNet mean/std) for the RGB image).

The arithmetic lives in third-party code that is not vendored in the reference tree:
  * torchvision==0.14.1 (requirements.txt:148): transforms.Resize on a PIL image calls ``img.resize(size[::-1], BILINEAR)``;
    ToTensor = uint8 HWC -> float32 CHW / 255; Normalize = (x - mean) / std in float32.
  * Pillow==9.3.0 (requirements.txt:92): Image.resize(BILINEAR) = ImagingResample (libImaging/Resample.c): separable triangle
    filter whose support is stretched by the down-scale factor (antialiasing), coefficients normalised in double, converted to
    22-bit fixed point, horizontal pass then vertical pass with a rounded uint8 intermediate.
Restated here in numpy from the published algorithm; pinned against the Pillow installed in this image (12.2.0, same resampling
code path) by tests/test_preprocess.py at test time and by the vectors in tests/golden/preprocess.npz."""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        arg = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        ww = 0.0
        for v in w:           # sequential double sum, as the C loop does
            ww += v
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        kk[xx, :xmax] = k
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along ``axis`` of an [H, W, C] uint8 image."""
    in_size = img.shape[axis]
    bounds, kk = _coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)            # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * kk[xx, x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_bilinear_u8(img: np.ndarray, size: int) -> np.ndarray:
    """Image.resize((size, size), BILINEAR) on an [H, W, C] (or [H, W]) uint8 array: horizontal pass, then vertical pass; a pass
    whose axis already has the target size is skipped (ImagingResample need_horizontal / need_vertical)."""
    squeeze = img.ndim == 2
    x = img[:, :, None] if squeeze else img
    if x.shape[1] != size:
        x = _resample_axis(x, size, axis=1)
    if x.shape[0] != size:
        x = _resample_axis(x, size, axis=0)
    return x[:, :, 0] if squeeze else x


def preprocess(img: np.ndarray, size: int, normalize: bool, flip: bool = False) -> np.ndarray:
    """uint8 [H, W, C] / [H, W] -> float32 [C, size, size]: hflip, resize, ToTensor, optional Normalize (sod_train.py:31-46)."""
    x = img[:, ::-1] if flip else img
    x = resize_bilinear_u8(np.ascontiguousarray(x), size)
    if x.ndim == 2:
        x = x[:, :, None]
    t = x.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    if normalize:
        mean = np.asarray(IMAGENET_MEAN, np.float32)[:, None, None]
        std = np.asarray(IMAGENET_STD, np.float32)[:, None, None]
        t = (t - mean) / std
    return t.astype(np.float32)
