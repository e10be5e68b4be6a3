"""AdamW with the reference's per-prefix learning-rate multipliers (config/sod.yml:56-76, config/cod.yml):
mmengine's ``paramwise_cfg.custom_keys`` semantics = the LONGEST matching substring key wins;
``bypass_duplicate`` = a parameter reachable under several names (the shared PReLU) is added once."""
from __future__ import annotations

from typing import Dict, Optional

import torch

SOD_CUSTOM_KEYS: Dict[str, float] = {
    "hitnet.backbone": 0.2,
    "hitnet.backbone.prompt_encoder.encoder2.downsample_layers": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.0": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.1": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.2": 0.02,
    "hitnet.backbone.prompt_encoder.encoder2.stages.3": 0.02,
}


def lr_mult_for(name: str, custom_keys: Dict[str, float]) -> float:
    best, mult = -1, 1.0
    for key, m in custom_keys.items():
        if key in name and len(key) > best:
            best, mult = len(key), m
    return mult


def build_optimizer(model: torch.nn.Module, lr: float = 5e-4, weight_decay: float = 0.1,
                    custom_keys: Optional[Dict[str, float]] = None, fused: Optional[bool] = None,
                    capturable: bool = False):
    custom_keys = SOD_CUSTOM_KEYS if custom_keys is None else custom_keys
    groups: Dict[float, list] = {}
    seen = set()
    for name, p in model.named_parameters(remove_duplicate=False):
        if id(p) in seen:   # bypass_duplicate; masters whose compute runs on a working copy have requires_grad=False
            continue
        seen.add(id(p))
        groups.setdefault(lr_mult_for(name, custom_keys), []).append(p)
    param_groups = [{"params": ps, "lr": lr * m, "initial_lr": lr * m} for m, ps in sorted(groups.items(), reverse=True)]
    if fused is None:
        fused = any(p.is_cuda for g in param_groups for p in g["params"])
    return torch.optim.AdamW(param_groups, lr=lr, weight_decay=weight_decay, fused=fused, capturable=capturable and fused)



class FlatAdamW:
    """AdamW over the flat buckets of a ``dist.GradReducer``: one ``dgtd_adamw_flat`` launch per run of equal learning rate per
    bucket (≈ 20 launches for 114 M parameters) that also rewrites the bf16 working copies, instead of torch's multi-tensor AdamW
    plus one cast per bucket.  Same update rule and defaults as ``torch.optim.AdamW``; ``param_groups`` carries one entry per
    distinct lr multiplier so the reference's epoch-wise cosine schedule (config/sod.yml:78-83) drives it unchanged."""

    def __init__(self, reducer, lr: float = 5e-4, weight_decay: float = 0.1, betas=(0.9, 0.999), eps: float = 1e-8,
                 custom_keys: Optional[Dict[str, float]] = None):
        from .. import _lib as L
        self._L = L
        self.reducer = reducer
        custom_keys = SOD_CUSTOM_KEYS if custom_keys is None else custom_keys
        self.betas, self.eps, self.weight_decay, self.steps = betas, eps, weight_decay, 0
        mults = sorted({lr_mult_for(n, custom_keys) for b in reducer.buckets for n in b["names"]}, reverse=True)
        self.param_groups = [{"lr": lr * m, "initial_lr": lr * m, "mult": m, "weight_decay": weight_decay} for m in mults]
        gidx = {m: i for i, m in enumerate(mults)}
        self.runs, self.state = [], []
        for b in reducer.buckets:
            if not b["mflat"].is_cuda:
                raise RuntimeError("FlatAdamW runs on the HIP device; use build_optimizer() (torch.optim.AdamW) on CPU")
            runs, off = [], 0
            for name, n in zip(b["names"], b["sizes"]):
                g = gidx[lr_mult_for(name, custom_keys)]
                # a run = contiguous elements with one lr that lie on one side of the working-copy boundary
                if runs and runs[-1][2] == g and not (off == b["n_work"]):
                    runs[-1][1] = off + n
                else:
                    runs.append([off, off + n, g])
                off += n
            self.runs.append(runs)
            self.state.append({"exp_avg": torch.zeros_like(b["mflat"]), "exp_avg_sq": torch.zeros_like(b["mflat"])})

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.reducer.zero_grad()

    @torch.no_grad()
    def step(self) -> None:
        L = self._L
        self.steps += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.steps, 1.0 - b2 ** self.steps
        st = L.stream_ptr()
        for b, runs, state in zip(self.reducer.buckets, self.runs, self.state):
            p, g, m, v, w, nw = b["mflat"], b["flat"], state["exp_avg"], state["exp_avg_sq"], b["wflat"], b["n_work"]
            for lo, hi, gi in runs:
                wp = (w.data_ptr() + 2 * lo) if (w is not None and hi <= nw) else None
                L.call("dgtd_adamw_flat", p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, wp,
                       hi - lo, float(self.param_groups[gi]["lr"]), b1, b2, self.eps, float(self.param_groups[gi]["weight_decay"]), bc1, bc2, st)

    def state_dict(self) -> dict:
        return {"steps": self.steps, "param_groups": [dict(g) for g in self.param_groups],
                "exp_avg": [s["exp_avg"].clone() for s in self.state], "exp_avg_sq": [s["exp_avg_sq"].clone() for s in self.state]}

    def load_state_dict(self, sd: dict) -> None:
        self.steps = int(sd["steps"])
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update(src)
        for s, a, b in zip(self.state, sd["exp_avg"], sd["exp_avg_sq"]):
            s["exp_avg"].copy_(a)
            s["exp_avg_sq"].copy_(b)
