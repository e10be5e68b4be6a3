#!/usr/bin/env python
"""bench.py — training-step throughput of the HIP-backed `cod` model on synthetic 512x512 RGB-D batches.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = forward + loss + backward + gradient all-reduce (RCCL, N>1) + AdamW update, batch 8 per GPU
(BASELINE.json configs[1]: config/sod.yml model, 512x512, batch 8, bf16).  Rank 0 prints ONE JSON line.
Inputs are resident in HBM before the timed region.  After the timed region a separate instrumented pass times
every C-ABI call with HIP events on its launch stream inside libdgtd.so (roofline; attention MFMA utilisation), and
rank 0 at N=1 reports the mIoU parity pair (HIP vs CPU oracle) and times the CPU oracle on a bounded sample (cpu_baseline).
Other BASELINE configs: --backbone pvt_v2_b3 (config 3's per-GPU workload), --mode predict --size 1024 --batch 4 --dtype f32
(config 4), --dtype f16 --batch 16 (config 5's per-GPU workload: fp16 + fp32 masters + dynamic loss scale).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")   # before torch loads MIOpen; see <package>/__init__.py (fp32 accuracy; the bf16 step has no Winograd-eligible library conv)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"hbm": (8000.0, "GB/s"), "mfma_bf16": (2500.0, "TFLOP/s"), "mfma_f16": (2500.0, "TFLOP/s"), "mfma_f32": (157.3, "TFLOP/s")}  # MI355X_MICROARCH.md
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="bf16: throughput mode; f16: the reference's AMP recipe (fp16 + fp32 masters + dynamic loss scale, BASELINE config 5); f32: parity mode")
    ap.add_argument("--mode", default="train", choices=["train", "predict"],
                    help="predict: eval-mode inference throughput (BASELINE config 4: --mode predict --size 1024 --batch 4 --dtype f32)")
    ap.add_argument("--backbone", default="pvt_v2_b2", help="pvt_v2_b3 = the tier-B backbone of BASELINE config 3")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="train mode: replay the step as a captured hipGraph (auto: try, fall back to the eager step if capture fails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-miou", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--all-kernels", default="", help="write EVERY kernel record of the instrumented pass (the JSON line keeps the top 12) to this file")
    return ap.parse_args()


def cpu_baseline(size: int, batch: int, mode: str, log=lambda m: None):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload: the same batch size, one
    untimed step, then at least two timed steps / 15 s."""
    from oracle import cod_cpu
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))   # the GPU box grants a 16-core share per GPU; more threads only oversubscribe
    torch.set_num_threads(cores)
    B = batch
    net = cod_cpu.cod(size)
    net = net.train() if mode == "train" else net.eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g)
    d = torch.rand(B, 1, size, size, generator=g)
    l = (torch.rand(B, 1, size, size, generator=g) > 0.5).float()

    def one():
        if mode == "train":
            net.zero_grad(set_to_none=True)
            net(None, x, l, d, mode="loss")["loss"].backward()
        else:
            with torch.no_grad():
                net(None, x, l, d, mode="predict")
    one()                                                    # untimed: allocator, oneDNN primitive caches
    steps, t0 = 0, time.perf_counter()
    while steps < 2 or time.perf_counter() - t0 < 15.0:      # ~15-30 s of CPU work
        one()
        steps += 1
        log(f"cpu oracle step {steps}: {time.perf_counter() - t0:.1f} s")
    dt = time.perf_counter() - t0
    what = "fwd+loss+bwd" if mode == "train" else "eval forward (predict)"
    return {"value": steps * B / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps {what} of oracle/cod_cpu.py, {size}x{size}, batch {B}, fp32, {cores} torch threads, {dt:.1f} s (after 1 untimed step)"}


def miou_parity(dgtd, dev, n: int = 256, size: int = 64, bs: int = 32, log=lambda m: None):
    """north_star "mIoU within 0.1 of reference on identical weights" (SURVEY 8(d)): binary 2-class mIoU of (sigmoid > 0.5) vs the
    label by ONE routine on the CPU oracle's and the HIP path's predict outputs over the same n synthetic samples with the
    deterministic filler weights, plus the reference's own mean_iou (twig/metric/mIOU.py:32-58) restated: identically 1.0 for the
    single-channel head on both sides.  Checker use of the oracle, outside the timed region."""
    from oracle import cod_cpu, filler
    M = dgtd.runner.metrics
    ref = cod_cpu.cod(size).eval()
    filler.fill_module(ref)
    net = dgtd.nn.cod(drop_path_rate=0.0)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev).eval()
    a, b, ra, rb = [], [], [], []
    t0 = time.perf_counter()
    for i in range(0, n, bs):
        x, d, l = filler.synthetic_batch(bs, size, seed=5000 + i)
        with torch.no_grad():
            pw, _ = ref(None, x, l, d, mode="predict")
            pg, _ = net(None, x.to(dev), l.to(dev), d.to(dev), mode="predict")
        pg = pg.cpu()
        for j in range(bs):
            a.append(M.binary_miou(pw[j:j + 1], l[j:j + 1]))
            b.append(M.binary_miou(pg[j:j + 1], l[j:j + 1]))
        ra.append(M.mean_iou_reference(pw, l))
        rb.append(M.mean_iou_reference(pg, l))
        log(f"mIoU parity: {i + bs}/{n} samples, {time.perf_counter() - t0:.1f} s")
    ma, mb = 100.0 * sum(a) / len(a), 100.0 * sum(b) / len(b)
    return {"binary_miou_oracle": round(ma, 4), "binary_miou_hip": round(mb, 4), "abs_diff_points": round(abs(ma - mb), 4),
            "reference_mean_iou_oracle": sum(ra) / len(ra), "reference_mean_iou_hip": sum(rb) / len(rb), "samples": n,
            "what": f"{n} synthetic {size}x{size} samples, filler weights, fp32 parity mode, predict path (cod.py:152-153, :219)"}


def main():
    args = parse()
    # stdout carries exactly ONE line, the result JSON: keep a private handle to it and point fd 1 at stderr so that library
    # banners (RCCL prints "Hostname / Librccl path" on stdout when the first communicator is created) cannot get in front of it
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)  # a stuck run shows where it is stuck
    t_start = time.perf_counter()
    try:   # no-op when libdgtd.so / libdgtd_torch.so are newer than their sources (they travel with the repo snapshot)
        import __graft_entry__ as _ge
        if int(os.environ.get("LOCAL_RANK", 0)) == 0:
            _ge.build()
        else:
            _ge.wait_for_build()     # local rank 0 (re)builds; the others only load finished libraries
    except Exception as e:  # keep going with the libraries already in the tree; a missing library still fails loudly below
        print(f"[bench] build() skipped: {e}", file=sys.stderr, flush=True)
    import dgtd
    rank, local, world = dgtd.dist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = DTYPES[args.dtype]
    train = args.mode == "train"

    torch.manual_seed(0)
    # random init of the reference architecture; DropPath active in train mode
    net = dgtd.nn.cod(compute_dtype=dtype, backbone=args.backbone).to(dev)
    net = net.train() if train else net.eval()
    dgtd.dist.broadcast_parameters(net)
    reducer = dgtd.dist.GradReducer(net, working_dtype=dtype, bucket_bytes=int(os.environ.get("DGTD_BUCKET_MB", "64")) << 20)
    flat_opt = os.environ.get("DGTD_FLAT_ADAMW", "1") != "0"
    scaler = dgtd.runner.LossScaler(dev) if (train and dtype == torch.float16) else None   # AmpOptimWrapper's GradScaler (config/sod.yml:57)
    # AdamW over the reducer's flat buckets (one launch per lr run, 16-bit working copies rewritten in the same pass), or torch's
    # fused multi-tensor AdamW + one cast per bucket
    want_graph = train and args.graph != "off" and os.environ.get("DGTD_GRAPH", "1") != "0" and (flat_opt or scaler is not None)
    opt = dgtd.runner.FlatAdamW(reducer, scaler=scaler, graph_safe=want_graph) if (flat_opt or scaler is not None) else dgtd.runner.build_optimizer(net)
    data = dgtd.runner.SyntheticRGBD(args.size, args.batch, rank=rank, device=dev)
    batches = [data.batch_at(i) for i in range(2)]  # resident in HBM before timing

    def train_step(i):
        b = batches[i % len(batches)]
        if (world > 1 or reducer._force) and not reducer.staged:
            reducer.set_staged(True)    # eager N > 1: complete buckets are all-reduced at the model's flush points, beside the rest of backward
        reducer.zero_grad()
        loss = net(b["raw"], b["input"], b["label"], b["depth"], mode="loss")["loss"]
        (scaler.scale(loss) if scaler is not None else loss).backward()
        reducer.finish()
        opt.step()
        if not isinstance(opt, dgtd.runner.FlatAdamW):
            reducer.refresh_working()
        return loss

    def predict_step(i):
        b = batches[i % len(batches)]
        with torch.no_grad():
            prob, _ = net(b["raw"], b["input"], b["label"], b["depth"], mode="predict")
        return prob.mean()

    step = train_step if train else predict_step
    graphed, graph_mode = None, None
    if want_graph:
        # N > 1: first the ONE-graph form with the RCCL all-reduces captured on the side stream (overlapped with backward), then the
        # split form (graph A | eager all-reduce | graph B), then the eager step with hook-driven overlap.  Every rank must take the
        # same path (the collectives differ), so the outcome of each attempt is agreed on with a MIN all-reduce.
        multi = world > 1 or reducer._force
        tried = set()
        for comm in ((os.environ.get("DGTD_GRAPH_COMM", "auto"), "split") if multi else ("auto",)):
            cand, ok, err = None, 1, None
            try:
                cand = dgtd.runner.GraphedTrainStep(net, reducer, opt, scaler=scaler, warmup=2, comm=comm)
                if cand.mode in tried:          # "auto" already resolved to this mode and failed
                    continue
                tried.add(cand.mode)
                tg = time.perf_counter()
                cand.capture(batches[0])
            except Exception as e:  # the eager step stays available; say so in the result line
                ok, err = 0, e
            if world > 1:
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                if ok and not int(flag.item()):
                    ok, err = 0, RuntimeError("capture failed on another rank")
            if ok:
                graphed, graph_mode = cand, cand.mode
                step = lambda i: graphed(batches[i % len(batches)])
                print(f"[bench] hipGraph capture of the training step ({cand.mode}): ok ({time.perf_counter() - tg:.1f} s)", file=sys.stderr, flush=True)
                break
            if cand is not None:
                cand.release()
            print(f"[bench] hipGraph capture ({comm}) failed ({type(err).__name__}: {err})", file=sys.stderr, flush=True)
        if graphed is None:
            if args.graph == "on":
                raise RuntimeError("hipGraph capture failed and --graph on was given")
            print("[bench] running the eager step", file=sys.stderr, flush=True)
            step = train_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        tw = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        log(f"warmup step {i}: {time.perf_counter() - tw:.2f} s")
    barrier()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(args.steps):
        th = time.perf_counter()
        loss = step(i)
        host += time.perf_counter() - th
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps: {dt:.2f} s")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    if final_loss != final_loss or final_loss in (float("inf"), float("-inf")):
        # random init at the config's lr 5e-4 (meant for pretrained weights) can diverge on the synthetic batches; NaN is not JSON
        print(f"[bench] final loss is not finite ({final_loss})", file=sys.stderr, flush=True)
        final_loss = None

    # ---- instrumented pass: per-call HIP-event timing inside libdgtd.so (not part of `value`)
    roofline, kernels, attention, entries = None, [], [], []
    if args.profile_steps > 0:
        # EVERY rank runs the instrumented steps (they are eager training steps: with N > 1 each issues the bucketed all-reduces, which
        # must meet their peers - rank 0 alone would hang against the others' barrier); rank 0 summarises.
        # The timing lives in the C ABI (dgtd_profile_enable), so this pass runs EXACTLY the autograd nodes of the timed step (the
        # C++ bindings and their fused nodes included).  The step is partly host-bound, so an event pair around a launch would also
        # time the host's enqueue gap: each instrumented step is queued BEHIND a ballast of large GEMMs (about 2.5 step-times of
        # device work), the host runs ahead, the launches and their event markers execute back to back, and the pairs measure
        # device time.
        ball = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        ball_out = torch.empty_like(ball)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(ball, ball, out=ball_out)
        e0.record()
        for _ in range(4):
            torch.mm(ball, ball, out=ball_out)
        e1.record()
        torch.cuda.synchronize()
        n_ball = int(2.5 * (1e3 * dt / args.steps) / max(e0.elapsed_time(e1) / 4, 0.1)) + 1
        dgtd._lib.profile_native(True)
        prof_step = train_step if train else predict_step      # the eager step: a graph replay does not pass through the C ABI
        for i in range(args.profile_steps):
            for _ in range(n_ball):
                torch.mm(ball, ball, out=ball_out)
            for _ in range(16):                                # empty brackets behind the same ballast: the event pair's own floor
                dgtd._lib.call("dgtd_profile_empty", dgtd._lib.stream_ptr())
            prof_step(i)
        del ball, ball_out
        summ = dgtd._lib.profile_native_summary()
        dgtd._lib.profile_native(False)
        log("instrumented pass done")
    if rank == 0 and args.profile_steps > 0:
        mfma_peak = {torch.bfloat16: "mfma_bf16", torch.float16: "mfma_f16", torch.float32: "mfma_f32"}[dtype]
        empty = summ.pop("dgtd_profile_empty", None)
        floor_ms = (empty["ms"] / empty["calls"]) if empty and empty["calls"] else 0.0      # per-call floor of the event bracket itself
        clock = (f"HIP event pair inside libdgtd.so around the entry's launches, minus the empty-bracket floor measured in the same pass "
                 f"({1e3 * floor_ms:.2f} us per call)")

        def record(key, calls, ms, amount, mfma):
            ms = max(ms - floor_ms * calls, 1e-6)
            peak, unit = PEAK[mfma_peak if mfma else "hbm"]
            per_s = amount / (ms * 1e-3)
            achieved = per_s / 1e12 if mfma else per_s / 1e9
            rec = {"kernel": key, "bound": "mfma" if mfma else "hbm", "achieved": round(achieved, 3), "peak": peak,
                   "unit": unit, "frac": round(achieved / peak, 4), "calls_per_step": calls / args.profile_steps,
                   "avg_us": round(1e3 * ms / calls, 2), "ms_per_step": round(ms / args.profile_steps, 3), "traffic": None}
            rec["algorithmic_flops" if mfma else "algorithmic_bytes"] = round(amount / calls)
            return rec

        by_entry = {}
        for key, e in summ.items():
            if e["amount"] <= 0 or e["ms"] <= 0:
                continue
            mfma = e["bound"] == "mfma"
            kernels.append(record(key, e["calls"], e["ms"], e["amount"], mfma))
            g = by_entry.setdefault((key.split("[")[0], mfma), {"calls": 0, "ms": 0.0, "amount": 0.0, "shapes": 0})
            g["calls"] += e["calls"]
            g["ms"] += e["ms"]
            g["amount"] += e["amount"]
            g["shapes"] += 1
        # HBM bytes per launch measured OFFLINE with rocprofv3 PMC passes of the same kernels at the same shapes (tools/pmc_run.sh:
        # FETCH_SIZE x2 + WRITE_SIZE in separate runs, MI355X_MICROARCH.md), read from the committed profiles/; null where no pass exists
        pmc = {}
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_by_bench_key.json"))):
            try:
                pmc.update(json.load(open(f)))
            except (OSError, ValueError):
                pass
        for k in kernels:
            k["traffic"] = pmc.get(k["kernel"])
        kernels.sort(key=lambda k: -k["ms_per_step"])
        # per C-ABI ENTRY over all its shapes (amount-weighted: total algorithmic bytes or flops / total device time): the entry with
        # the most time per step is the step's dominant hand-written kernel and carries the bench line's `roofline`
        for (name, mfma), g in by_entry.items():
            r = record(name, g["calls"], g["ms"], g["amount"], mfma)
            r["shapes"] = g["shapes"]
            entries.append(r)
        entries.sort(key=lambda k: -k["ms_per_step"])
        # MFMA utilisation of the attention GEMMs (QK^T and AV, and their gradients), every stage: flops / device time / dense peak
        attention = [k for k in kernels if k["kernel"].startswith("dgtd_sra_attn")]
        if args.all_kernels:
            with open(args.all_kernels, "w") as f:
                json.dump({"entries": entries, "kernels": kernels, "clock": clock}, f, indent=0)
        if entries:
            top = entries[0]
            roofline = {k: top[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
            # traffic: PMC bytes of the entry's shapes that have a pass, scaled to the entry's average launch
            shapes = [k for k in kernels if k["kernel"].split("[")[0] == top["kernel"] and k["traffic"]]
            if shapes and not top["bound"] == "mfma":
                cov_alg = sum(k["algorithmic_bytes"] * k["calls_per_step"] for k in shapes)
                cov_pmc = sum(k["traffic"] * k["calls_per_step"] for k in shapes)
                roofline["traffic"] = round(top["algorithmic_bytes"] * cov_pmc / cov_alg)
                roofline["traffic_source"] = (f"offline rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE) for {len(shapes)} of the "
                                              f"entry's {top['shapes']} shapes: measured / algorithmic = {cov_pmc / cov_alg:.3f}, applied to the average launch")
            else:
                roofline["traffic"] = None
            for extra in ("algorithmic_bytes", "algorithmic_flops"):
                if extra in top:
                    roofline[extra] = top[extra]
            roofline.update(kernel=top["kernel"], avg_us=top["avg_us"], ms_per_step=top["ms_per_step"], calls_per_step=top["calls_per_step"],
                            shapes=top["shapes"], clock=clock,
                            what="the C-ABI entry with the most device time per step, all its shapes pooled (amount-weighted)")
    if world > 1:
        torch.distributed.barrier()

    if rank == 0:
        imgs = args.steps * args.batch * world
        what = "fwd+loss+bwd+allreduce+AdamW" if train else "eval forward, predict mode"
        flops_per_img = (786.7e9 if train else 262.2e9) * (args.size / 512) ** 2      # reference-algorithmic, pvt_v2_b2
        out = {
            "metric": (f"training images/sec ({args.size}x{args.size} RGB-D, bs={args.batch} per GPU; {what})" if train
                       else f"inference images/sec ({args.size}x{args.size} RGB-D, bs={args.batch} per GPU; {what})"),
            "value": round(imgs / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "per_gpu": round(imgs / dt / world, 3), "target_per_gpu": 40.0 if train else None,
            "config": {"workload": f"config/sod.yml model `cod` ({args.backbone} + ConvNeXt-B texture diffuser + Hitnet decoder), "
                                   f"{args.size}x{args.size} RGB+depth, batch {args.batch}/GPU, random init, "
                                   + ("DropPath active" if train else "eval mode, no_grad"),
                       "mode": args.mode, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "hip_graph": graphed is not None, "graph_mode": graph_mode,
                       "allreduce_payload": (None if not (world > 1 or reducer._force) else ("16-bit working dtype + fp32 rest" if reducer.comm16 else "fp32")),
                       "final_loss": round(final_loss, 4) if (train and final_loss is not None) else None,
                       "host_enqueue_ms_per_step": round(1e3 * host / args.steps, 2),
                       "reference_algorithmic_tflops": round(imgs / dt * flops_per_img / 1e12, 2),
                       "loss_scale": scaler.get_scale() if scaler is not None else None},
            "roofline": roofline, "entries": entries[:10], "kernels": kernels[:12], "attention_mfma": attention,
        }
        if world == 1 and not args.no_miou:
            log("mIoU parity (HIP vs CPU oracle, 256 synthetic samples) ...")
            out["miou"] = miou_parity(dgtd, dev, log=log)
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on a bounded sample ...")
            del net, reducer, opt
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.size, args.batch, args.mode, log)
        faulthandler.cancel_dump_traceback_later()
        print(json.dumps(out), file=result_out, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
