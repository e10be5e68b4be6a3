"""Subprocess body of test_graphed_step_with_forced_allreduce (tests/test_train_gpu.py): a world-1 RCCL process group
(DGTD_FORCE_ALLREDUCE=1) so the N > 1 code paths of the captured step run on a one-GPU box: hook-driven gather + bucketed
all-reduce on the side stream (eager), the same captured INSIDE one hipGraph ("fused"), and graph A | all-reduce | graph B
("split").  Prints one JSON line: losses per mode and the largest weight difference to the eager run."""
import json
import os
import sys

os.environ["DGTD_FORCE_ALLREDUCE"] = "1"
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
os.environ.setdefault("DGTD_GEMM_CANDIDATES", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dgtd  # noqa: E402
from oracle import filler  # noqa: E402


def main():
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1]]
    # fp32: every N > 1 form; bf16: the two a run can take (fused with RCCL, split otherwise) - "hooks" differs from "fused" only in
    # WHERE the gathers are issued, which the fp32 case pins to atomic noise
    MODES = ("fused", "hooks", "split") if sys.argv[1] == "f32" else ("fused", "split")
    os.environ.setdefault("MASTER_PORT", sys.argv[2])
    rank, local, world = dgtd.dist.init_process_group()
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    S, B = 64, 2
    data = dgtd.runner.SyntheticRGBD(S, B, device="cuda")
    batches = [data.batch_at(i) for i in range(3)]

    import copy
    torch.manual_seed(0)
    base = dgtd.nn.cod(drop_path_rate=0.0, compute_dtype=dtype)      # built and filled ONCE (5 + 2..20 s); every mode starts from a copy
    filler.fill_module(base)

    def make():
        net = copy.deepcopy(base).cuda().train()
        red = dgtd.dist.GradReducer(net, bucket_bytes=16 << 20, working_dtype=dtype)
        assert red._force and red.overlap and red.comm_stream is not None and len(red.buckets) >= 4
        assert red.comm16 == (dtype != torch.float32)
        opt = dgtd.runner.FlatAdamW(red, lr=1e-4, graph_safe=True)
        return net, red, opt

    out = {}
    net_e, red_e, opt_e = make()
    losses = []
    for i in range(3):
        b = batches[i]
        red_e.zero_grad()
        loss = net_e(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
        loss.backward()
        red_e.finish()
        opt_e.sync_lr()
        opt_e.step()
        losses.append(loss.item())
    out["eager"] = {"losses": losses}
    print("eager done", flush=True)
    for mode in MODES:
        net_g, red_g, opt_g = make()
        stepper = dgtd.runner.GraphedTrainStep(net_g, red_g, opt_g, warmup=1, comm=mode)
        stepper.capture(batches[0])
        print(f"{mode}: captured", flush=True)
        assert stepper.mode == mode and (stepper.graph_opt is not None) == (mode == "split")
        assert red_g.overlap, "the reducer's own mode is restored after capture"
        losses = [stepper(batches[i]).item() for i in range(3)]
        torch.cuda.synchronize()
        worst, bad = 0.0, 0
        for (k, p), (_, q) in zip(net_g.named_parameters(), net_e.named_parameters()):
            d = (p - q).abs()
            worst = max(worst, float(d.max()))
            bad += int((d > 2e-5 + 1e-3 * q.abs()).sum())
        out[mode] = {"losses": losses, "max_weight_diff": worst, "elements_off": bad, "steps": opt_g.steps}
        print(f"{mode}: replayed", flush=True)
        stepper.release()
        del net_g, red_g, opt_g, stepper
        torch.cuda.synchronize()
    print("RESULT " + json.dumps(out), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
