"""TEST INFRASTRUCTURE — deterministic, name-keyed weight filler (SURVEY.md §8(c) "Weights for goldens").

458 MB of fp32 weights cannot be committed, so every golden vector is generated with weights
that are a pure function of (state_dict key, element index).  Integer arithmetic only
(crc32 -> splitmix64 counter stream -> Box–Muller in fp64), independent of torch's RNG and
version, so the same weights can be rebuilt anywhere: in the build container on the
reference, and on the GPU box on the oracle restatement and on the HIP-backed modules.
"""
from __future__ import annotations

import json
import os
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _uniform(key: str, n: int, stream: int) -> np.ndarray:
    """n doubles in (0,1), a pure function of (key, stream, index)."""
    seed = np.uint64((zlib.crc32(key.encode("utf-8")) * 0x100000001B3 + stream) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed) & _MASK
    bits = _splitmix64(_splitmix64(ctr)) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(key: str, n: int) -> np.ndarray:
    u1 = _uniform(key, n, 1)
    u2 = _uniform(key, n, 2)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def uniform(key: str, n: int) -> np.ndarray:
    return _uniform(key, n, 3)


_CACHE = {}   # (key, shape) -> fp32 tensor: a test session fills ~25 models with the same 114 M values (13 - 22 s each to generate)

# The same values for the processes a test SPAWNS (the 2-rank tests, the forced-all-reduce workers): one flat fp32 file + an index in
# shared memory, written by the parent (save_disk_cache) and memory-mapped by the children.  The file name carries a checksum of this
# source file, so a change of the rules below can never meet stale values.
_DISK = None          # (flat fp32 memmap, {"key|shape": [offset, n]}) or False when there is no file
_NEW = {}             # entries generated in this process that the file does not hold


def _disk_paths():
    import tempfile
    with open(__file__, "rb") as f:
        tag = zlib.crc32(f.read())
    base = os.environ.get("DGTD_FILLER_CACHE_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    stem = os.path.join(base, f"dgtd_filler_{os.getuid()}_{tag:08x}")
    return stem + ".npy", stem + ".json"


def _disk():
    global _DISK
    if _DISK is None:
        npy, idx = _disk_paths()
        try:
            with open(idx) as f:
                index = json.load(f)
            _DISK = (np.load(npy, mmap_mode="r"), index)
        except (OSError, ValueError):
            _DISK = False
    return _DISK


def save_disk_cache() -> bool:
    """Write every fp32 value this process holds (generated or read) to the shared file; False when nothing new would be added."""
    global _DISK
    if not _NEW:
        return False
    npy, idx = _disk_paths()
    index, parts, off = {}, [], 0
    for (key, shape), t in _CACHE.items():
        a = t.detach().reshape(-1).numpy()
        index[key + "|" + ",".join(map(str, shape))] = [off, int(a.size)]
        parts.append(a)
        off += int(a.size)
    flat = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    tmp = f"{npy}.{os.getpid()}.tmp.npy"
    np.save(tmp, flat)
    os.replace(tmp, npy)
    with open(f"{idx}.{os.getpid()}.tmp", "w") as f:
        json.dump(index, f)
    os.replace(f"{idx}.{os.getpid()}.tmp", idx)
    _NEW.clear()
    _DISK = None
    return True


def value_for(key: str, shape, dtype=torch.float32) -> torch.Tensor:
    """The filler's rule table, keyed on the reference's state_dict naming
    (twig/model/cod.py; key inventory in SURVEY.md §2.2)."""
    if dtype == torch.float32:
        shape = tuple(shape)
        hit = _CACHE.get((key, shape))
        if hit is None:
            d = _disk()
            ent = d[1].get(key + "|" + ",".join(map(str, shape))) if d else None
            if ent is not None:
                hit = torch.from_numpy(np.array(d[0][ent[0]:ent[0] + ent[1]], dtype=np.float32)).reshape(shape)
            else:
                hit = _value_for(key, shape, dtype)
                _NEW[(key, shape)] = True
            _CACHE[(key, shape)] = hit
        return hit
    return _value_for(key, shape, dtype)


def _value_for(key: str, shape, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        v = 0.1 * normal(key, n)
    elif leaf == "running_var":
        v = 0.5 + uniform(key, n)
    elif leaf == "gamma":  # ConvNeXt layer scale; reference constructs it at 1.0 (cod.py:1144)
        v = 0.5 + 0.1 * normal(key, n)
    elif leaf == "bias":
        v = 0.05 * normal(key, n)
    elif leaf == "weight" and len(shape) == 1:
        if n == 1:  # the shared nn.PReLU() slope (cod.py:686)
            v = np.full(1, 0.25)
        else:  # LayerNorm / BatchNorm scale
            v = 1.0 + 0.1 * normal(key, n)
    else:  # conv / linear kernels: LeCun-scaled so activations stay O(1) through the net
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        v = normal(key, n) / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(np.asarray(v, dtype=np.float64).reshape(shape)).to(dtype)


@torch.no_grad()
def fill_module(module: torch.nn.Module, prefix: str = "") -> None:
    """Overwrite every parameter and buffer of ``module`` in place.  ``prefix`` is prepended to
    the local key so a sub-module can be filled with the values it has inside the full model
    (e.g. prefix='hitnet.backbone.block1.0.attn.')."""
    sd = module.state_dict()
    for k, t in sd.items():
        t.copy_(value_for(prefix + k, tuple(t.shape), t.dtype if t.is_floating_point() else torch.float32).to(t.dtype))


def synthetic_batch(batch: int, size: int, seed: int = 1234):
    """Deterministic inputs with the dataset's dict contract (twig/dataset/sod_train.py:55-83):
    input ~ N(0,1) [B,3,S,S]; depth smooth in [0,1] [B,1,S,S]; label binary blobs [B,1,S,S]."""
    key = f"synthetic/{seed}/{batch}/{size}"
    x = normal(key + "/input", batch * 3 * size * size).reshape(batch, 3, size, size)
    lo = max(size // 16, 2)
    d_lo = torch.from_numpy(uniform(key + "/depth", batch * lo * lo).reshape(batch, 1, lo, lo)).float()
    depth = torch.nn.functional.interpolate(d_lo, size=(size, size), mode="bilinear", align_corners=False)
    l_lo = torch.from_numpy(uniform(key + "/label", batch * lo * lo).reshape(batch, 1, lo, lo)).float()
    label = (torch.nn.functional.interpolate(l_lo, size=(size, size), mode="bilinear", align_corners=False) > 0.55).float()
    return torch.from_numpy(x).float(), depth.clamp_(0, 1), label
