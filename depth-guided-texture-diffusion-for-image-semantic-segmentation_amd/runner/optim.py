"""AdamW with the reference's per-prefix learning-rate multipliers (config/sod.yml:56-76, config/cod.yml):
mmengine's ``paramwise_cfg.custom_keys`` semantics = the LONGEST matching substring key wins;
``bypass_duplicate`` = a parameter reachable under several names (the shared PReLU) is added once."""
from __future__ import annotations

import os
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



class LossScaler:
    """Dynamic loss scaling with torch.amp.GradScaler's rule (the reference's AmpOptimWrapper, config/sod.yml:57): the loss is
    multiplied by ``scale``; a step whose gradients hold an inf / NaN is skipped and halves the scale, ``growth_interval`` clean
    steps in a row double it.  All state lives on the device (``state`` = [scale, growth_tracker, 1/scale, found_inf, optimizer
    steps actually taken]) and is updated by ``dgtd_loss_scale_update``: no host synchronisation per step."""

    def __init__(self, device, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, int(growth_interval)
        self.state = torch.tensor([init_scale, 0.0, 1.0 / init_scale, 0.0, 0.0], dtype=torch.float32, device=device)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self.state[0]

    def get_scale(self) -> float:
        return float(self.state[0].item())

    def state_dict(self) -> dict:
        st = self.state.tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[1]), "steps_taken": int(st[4])}

    def load_state_dict(self, sd: dict) -> None:
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state.copy_(torch.tensor([sd["scale"], float(sd["_growth_tracker"]), 1.0 / sd["scale"], 0.0, float(sd.get("steps_taken", 0))]))


_ADAM_G16 = os.environ.get("DGTD_ADAM_G16", "1") != "0"      # 0: always widen the 16-bit all-reduce payload into the fp32 bucket first


class FlatAdamW:
    """AdamW over the flat buckets of a ``dist.GradReducer``: one ``dgtd_adamw_flat`` launch per run of equal learning rate per
    bucket (≈ 20 launches for 114 M parameters) that also rewrites the 16-bit working copies, instead of torch's multi-tensor AdamW
    plus one cast per bucket.  Same update rule and defaults as ``torch.optim.AdamW``; ``param_groups`` carries one entry per
    distinct lr multiplier so the reference's epoch-wise cosine schedule (config/sod.yml:78-83) drives it unchanged.
    ``scaler``: a ``LossScaler`` (fp16 mode): the gradients are checked for inf / NaN (one pass per bucket), un-scaled inside
    the AdamW launch, and an overflowed step is skipped on the device.
    ``state_dict()`` has torch.optim.AdamW's layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in the parameters'
    logical shapes, ``param_groups`` with parameter indices, plus ``param_names``), whatever the bucket size was."""

    def __init__(self, reducer, lr: float = 5e-4, weight_decay: float = 0.1, betas=(0.9, 0.999), eps: float = 1e-8,
                 custom_keys: Optional[Dict[str, float]] = None, scaler: Optional[LossScaler] = None, graph_safe: bool = False):
        from .. import _lib as L
        self._L = L
        self.reducer, self.scaler = reducer, scaler
        custom_keys = SOD_CUSTOM_KEYS if custom_keys is None else custom_keys
        self.betas, self.eps, self.weight_decay, self._steps = betas, eps, weight_decay, 0
        self.pipelined = False      # set by GraphedTrainStep(comm="fused"): every bucket's AdamW waits for that bucket's all-reduce only
        mults = sorted({lr_mult_for(n, custom_keys) for b in reducer.buckets for n in b["names"]}, reverse=True)
        self.param_groups = [{"lr": lr * m, "initial_lr": lr * m, "mult": m, "weight_decay": weight_decay} for m in mults]
        gidx = {m: i for i, m in enumerate(mults)}
        self.slots, self.runs, self.state = [], [], []
        for b in reducer.buckets:
            if not b["mflat"].is_cuda:
                raise RuntimeError("FlatAdamW runs on the HIP device; use build_optimizer() (torch.optim.AdamW) on CPU")
            # slot = (offset, padded length, group); the alignment padding after a tensor travels with it (zeros: a no-op update)
            slots = [(off, n, gidx[lr_mult_for(name, custom_keys)]) for name, off, n in zip(b["names"], b["offsets"], b["padded"])]
            self.slots.append(slots)
            self.runs.append(self._merge(slots, b["n_work"], ()))
            self.state.append({"exp_avg": torch.zeros_like(b["mflat"]), "exp_avg_sq": torch.zeros_like(b["mflat"])})
        wd = reducer.working_dtype
        self._w_dt = L.F16 if wd == torch.float16 else L.BF16
        # graph_safe: no per-step HOST value may enter a launch (a captured hipGraph would replay it frozen): the step count for the
        # bias corrections and the learning rates live in device memory.  Without a loss scaler the same device state is used with
        # scale = 1 and growth/backoff = 1, so only its step counter moves.
        self.graph_safe = graph_safe
        dev = reducer.buckets[0]["mflat"].device if reducer.buckets else "cpu"
        self._state = scaler.state if scaler is not None else (
            torch.tensor([1.0, 0.0, 1.0, 0.0, 0.0], dtype=torch.float32, device=dev) if graph_safe else None)
        self._lr_dev = torch.tensor([g["lr"] for g in self.param_groups], dtype=torch.float32, device=dev) if graph_safe else None
        self._lr_host = [g["lr"] for g in self.param_groups]

    @property
    def steps(self) -> int:
        """Optimizer steps actually taken (fp16 mode: read from the scaler's device state - steps it skipped do not count)."""
        return int(self._state[4].item()) if self._state is not None else self._steps

    def sync_lr(self) -> None:
        """graph_safe: push the param_groups' learning rates (what a scheduler writes) to device memory; a no-op while unchanged."""
        if self._lr_dev is not None:
            cur = [float(g["lr"]) for g in self.param_groups]
            if cur != self._lr_host:
                self._lr_dev.copy_(torch.tensor(cur, dtype=torch.float32))
                self._lr_host = cur

    @staticmethod
    def _merge(slots, n_work, missing):
        """Runs = maximal contiguous stretches with one lr on one side of the working-copy boundary, skipping ``missing`` slots."""
        runs, prev_end = [], None
        for i, (off, n, g) in enumerate(slots):
            if i in missing:
                prev_end = None
                continue
            if runs and prev_end == off and runs[-1][2] == g and off != n_work:
                runs[-1][1] = off + n
            else:
                runs.append([off, off + n, g])
            prev_end = off + n
        return runs

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.reducer.zero_grad()

    @torch.no_grad()
    def step(self) -> None:
        L = self._L
        self._steps += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self._steps, 1.0 - b2 ** self._steps
        st = L.stream_ptr()
        amp = self._state.data_ptr() if self._state is not None else None
        if self.scaler is not None:
            for b in self.reducer.buckets:          # GradScaler.unscale_'s found_inf over every (already all-reduced) gradient
                if self.pipelined:
                    self.reducer.wait_bucket(b["index"])
                L.call("dgtd_found_inf", b["flat"].data_ptr(), b["flat"].numel(), amp + 12, st)
        lrp = self._lr_dev.data_ptr() if self._lr_dev is not None else None
        for b, slots, runs, state in zip(self.reducer.buckets, self.slots, self.runs, self.state):
            # pipelined: the bucket's all-reduce runs on the reducer's side stream and is joined bucket by bucket; without a loss scaler
            # (no found_inf pass over the fp32 bucket) the working-copy segment is updated straight from the 16-bit payload
            g16 = b.get("g16") if (self.pipelined and self.scaler is None and b.get("works") and _ADAM_G16) else None
            if self.pipelined:
                self.reducer.wait_bucket(b["index"], widen=g16 is None)
            p, g, m, v, w, nw = b["mflat"], b["flat"], state["exp_avg"], state["exp_avg_sq"], b["wflat"], b["n_work"]
            # world 1: parameters without a gradient this step are skipped, like torch.optim.AdamW does.  N > 1: ``missing`` is
            # rank-local while the all-reduced slot holds the average over every rank (zero from the ranks that did not use the
            # parameter) - every rank applies the same update, so the replicas cannot drift apart
            if b["missing"] and self.reducer.world == 1:
                runs = self._merge(slots, nw, set(b["missing"]))
            for lo, hi, gi in runs:
                wp = (w.data_ptr() + 2 * lo) if (w is not None and hi <= nw) else None
                direct = g16 is not None and hi <= nw
                L.call("dgtd_adamw_flat_g16" if direct else "dgtd_adamw_flat_amp", p.data_ptr() + 4 * lo,
                       (g16.data_ptr() + 2 * lo) if direct else (g.data_ptr() + 4 * lo), m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, wp,
                       self._w_dt, hi - lo, float(self.param_groups[gi]["lr"]), b1, b2, self.eps,
                       float(self.param_groups[gi]["weight_decay"]), bc1, bc2, amp,
                       (lrp + 4 * gi) if lrp is not None else None, st)
        if self.scaler is not None:
            L.call("dgtd_loss_scale_update", self.scaler.state.data_ptr(), float(self.scaler.growth_factor),
                   float(self.scaler.backoff_factor), int(self.scaler.growth_interval), st)
        elif self._state is not None:               # only the device-side step counter moves
            L.call("dgtd_loss_scale_update", amp, 1.0, 1.0, 1 << 30, st)

    # ------------------------------------------------------------------ torch.optim.AdamW-compatible state
    def _logical(self, b, i, flat):
        off, n, shape = b["offsets"][i], b["sizes"][i], b["shapes"][i]
        t = flat[off:off + n]
        if b["nhwc"][i]:
            return t.view(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2)
        return t.view(shape)

    def state_dict(self) -> dict:
        steps = self.steps
        names, state, members = [], {}, [[] for _ in self.param_groups]
        for b, slots, st in zip(self.reducer.buckets, self.slots, self.state):
            for i, name in enumerate(b["names"]):
                idx = len(names)
                names.append(name)
                members[slots[i][2]].append(idx)
                state[idx] = {"step": torch.tensor(float(steps)),
                              "exp_avg": self._logical(b, i, st["exp_avg"]).detach().cpu().contiguous(),
                              "exp_avg_sq": self._logical(b, i, st["exp_avg_sq"]).detach().cpu().contiguous()}
        groups = [{"lr": g["lr"], "initial_lr": g["initial_lr"], "betas": tuple(self.betas), "eps": self.eps,
                   "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False, "params": members[i]}
                  for i, g in enumerate(self.param_groups)]
        out = {"state": state, "param_groups": groups, "param_names": names}
        if self.scaler is not None:
            out["loss_scaler"] = self.scaler.state_dict()
        return out

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        names = sd.get("param_names")
        index = {n: i for i, n in enumerate(names)} if names is not None else None
        steps, k = None, 0
        for b, st in zip(self.reducer.buckets, self.state):
            for i, name in enumerate(b["names"]):
                idx = index[name] if index is not None else k
                k += 1
                src = sd["state"][idx]
                if tuple(src["exp_avg"].shape) != b["shapes"][i]:
                    raise ValueError(f"optimizer state of {name}: shape {tuple(src['exp_avg'].shape)} != parameter shape {b['shapes'][i]}")
                self._logical(b, i, st["exp_avg"]).copy_(src["exp_avg"])
                self._logical(b, i, st["exp_avg_sq"]).copy_(src["exp_avg_sq"])
                steps = int(float(src["step"])) if steps is None else steps
        self._steps = steps or 0
        if self._state is not None:
            self._state[4] = float(self._steps)
        by_mult = sorted(sd["param_groups"], key=lambda g: -g["initial_lr"])
        for g, src in zip(self.param_groups, by_mult):
            g["lr"], g["initial_lr"], g["weight_decay"] = src["lr"], src["initial_lr"], src["weight_decay"]
        if self.scaler is not None and "loss_scaler" in sd:
            self.scaler.load_state_dict(sd["loss_scaler"])
