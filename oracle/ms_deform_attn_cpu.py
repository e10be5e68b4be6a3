"""TEST INFRASTRUCTURE — CPU restatement of the reference's multi-scale deformable attention
(twig/ops/functions/ms_deform_attn_func.py:49-71, ``ms_deform_attn_core_pytorch``: per level, F.grid_sample of the value map at
2*loc-1 with bilinear / zeros / align_corners=False, weighted by the attention weights and summed over levels and points).
Pinned against the reference's own function (imported in the build container by oracle/make_golden.py through a stub for the
compiled ``MultiScaleDeformableAttention`` module) with the vectors in tests/golden/msda.npz, and by the reference's acceptance
thresholds (twig/ops/test.py:43 allclose in double, :68 rtol 1e-2 / atol 1e-3 in float, :96-99 gradcheck)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    levels = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lid, (h, w) in enumerate(spatial_shapes):
        v = levels[lid].flatten(2).transpose(1, 2).reshape(N * M, D, int(h), int(w))           # [N*M, D, H, W]
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)                                    # [N*M, Lq, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))   # [N*M, D, Lq, P]
    a = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def case_inputs(name: str, N, M, D, Lq, shapes, P, dtype=torch.float64):
    """Deterministic inputs (oracle/filler streams, independent of torch's RNG), laid out like twig/ops/test.py:25-33."""
    from . import filler
    import numpy as np
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    u = lambda key, shape: torch.from_numpy(filler.uniform(f"msda/{name}/{key}", int(np.prod(shape))).reshape(shape)).to(dtype)
    value = u("value", (N, S, M, D)) * 0.01
    loc = u("loc", (N, Lq, M, L, P, 2)) * 1.2 - 0.1          # a few samples fall outside [0,1]: zero padding is exercised
    attn = u("attn", (N, Lq, M, L, P)) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    grad = u("grad", (N, Lq, M * D)) - 0.5
    return value, torch.tensor(shapes, dtype=torch.long), loc, attn, grad


CASES = {   # name: (N, M, D, Lq, shapes, P)
    "ref_test": (1, 2, 2, 2, [(6, 4), (3, 2)], 2),                  # the shapes of twig/ops/test.py:15-20
    "d30": (2, 2, 30, 5, [(6, 4), (3, 2)], 2),
    "d71": (1, 3, 71, 4, [(5, 7), (3, 3), (2, 2)], 3),
    "d64": (2, 8, 64, 50, [(16, 16), (8, 8), (4, 4), (2, 2)], 4),  # a Deformable-DETR-like layer
}
