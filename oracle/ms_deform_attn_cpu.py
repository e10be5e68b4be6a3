"""TEST INFRASTRUCTURE — CPU restatement of the reference's multi-scale deformable attention
(semantics of twig/ops/functions/ms_deform_attn_func.py:49-71, ``ms_deform_attn_core_pytorch``, and of the kernels behind
MSDA.ms_deform_attn_forward: bilinear sampling of each level's value map at the sampling locations, zero padding,
align_corners=False, weighted by the attention weights and summed over levels and points).
Pinned against the reference's own function (imported in the build container by oracle/make_golden.py through a stub for the
compiled ``MultiScaleDeformableAttention`` module) with the vectors in tests/golden/msda.npz, and by the reference's acceptance
thresholds (twig/ops/test.py:43 allclose in double, :68 rtol 1e-2 / atol 1e-3 in float, :96-99 gradcheck)."""
from __future__ import annotations

import torch


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """First-principles evaluation (explicit corner gathers) of what the reference computes with one F.grid_sample per level:
    pixel coordinates h = y*H - 0.5, w = x*W - 0.5 (align_corners=False), bilinear weights on the four neighbours, neighbours
    outside the map contribute zero (padding_mode='zeros').  Differentiable through torch ops, so autograd gives the oracle
    gradients w.r.t. value, locations and weights."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = value.new_zeros(N, Lq, M, D)
    start = 0
    for lvl, (H, W) in enumerate((int(h), int(w)) for h, w in spatial_shapes):
        maps = value[:, start:start + H * W].permute(0, 2, 1, 3)            # [N, M, H*W, D]
        start += H * W
        px = sampling_locations[:, :, :, lvl, :, 0] * W - 0.5                # [N, Lq, M, P]
        py = sampling_locations[:, :, :, lvl, :, 1] * H - 0.5
        x0, y0 = torch.floor(px), torch.floor(py)
        fx, fy = px - x0, py - y0
        a = attention_weights[:, :, :, lvl, :]                               # [N, Lq, M, P]
        for dy, dx, wgt in ((0, 0, (1 - fy) * (1 - fx)), (0, 1, (1 - fy) * fx), (1, 0, fy * (1 - fx)), (1, 1, fy * fx)):
            xi, yi = x0 + dx, y0 + dy
            inside = ((xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)).to(value.dtype)
            flat = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()      # [N, Lq, M, P]
            idx = flat.permute(0, 2, 1, 3).reshape(N, M, Lq * P, 1).expand(N, M, Lq * P, D)
            corner = torch.gather(maps, 2, idx).reshape(N, M, Lq, P, D).permute(0, 2, 1, 3, 4)    # [N, Lq, M, P, D]
            out = out + ((a * wgt * inside).unsqueeze(-1) * corner).sum(3)
    return out.reshape(N, Lq, M * D)


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
