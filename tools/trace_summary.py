#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV for ONE steady-state training step.

    python tools/trace_summary.py gpurun_out/prof/.../*_kernel_trace.csv [--top 40] > profiles/rNN_step_kernels.txt

Steps are delimited by the AdamW kernels (dgtd adamw_flat_kernel or torch's multi_tensor_apply); the LAST full step is reported so
MIOpen's first-step solver search and allocator warm-up are excluded."""
import collections
import csv
import sys


def category(n: str) -> str:
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
        return "library GEMM (hipBLASLt)"
    if any(k in n for k in ("igemm", "naive_conv", "SubTensorOp", "batched_transpose", "Im2d", "Col2Im", "grouped_conv", "gridwise", "miopen")):
        return "library dense conv (MIOpen, incl. its layout transforms)"
    if any(k in n for k in ("gemm_tn_kernel", "transpose_batched_kernel", "dkdv_reduce_kernel")):
        return "dgtd own GEMM (csrc/gemm.hip) + weight transposes"
    if any(k in n for k in ("dwconv", "ln_fwd", "ln_bwd", "sra_", "attn_delta", "diffus", "colsum", "scale_residual", "conv3x3", "bilinear_fwd_kernel", "bilinear_bwd_kernel",
                            "prelu_", "ca_gate", "ca_apply", "pooled_sum", "loss_", "im2col", "col2im", "multi_copy", "multi_reduce", "ssim_", "found_inf", "preprocess", "resize_", "bn_partial", "bn_apply", "bn_bwd", "sam_")):
        return "dgtd HIP kernels"
    if "reduce_kernel" in n:
        return "torch reductions"
    if "FusedOptim" in n or "multi_tensor" in n or "adamw_flat" in n:
        return "AdamW / multi-tensor"
    if "fill" in n.lower():
        return "fills / memsets"
    if "upsample" in n:
        return "bilinear resampling"
    if "avg_pool" in n:
        return "avg_pool (loss box filter)"
    if "batch_norm" in n:
        return "batch_norm"
    if "copy" in n.lower() or "Cat" in n:
        return "torch copies / casts / cat"
    if "elementwise" in n or "prelu" in n:
        return "torch elementwise"
    return "other (FFT, pad, roll, ...)"


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "adamw_flat_kernel" in r["Kernel_Name"]]
    groups, prev = [], None
    for i in opt:
        if prev is None or i - prev > 50:
            groups.append([i, i])
        else:
            groups[-1][1] = i
        prev = i
    skip = int(sys.argv[sys.argv.index("--skip-last") + 1]) if "--skip-last" in sys.argv else 0   # bench.py: 2 instrumented steps
    a, b = groups[-2 - skip][1] + 1, groups[-1 - skip][1] + 1
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b - 1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    busy = 0
    for r in rows[a:b]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += d
        agg[r["Kernel_Name"]][1] += 1
        busy += d
    print(f"# last full step: wall {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, {b - a} kernel launches")
    cats = collections.defaultdict(lambda: [0, 0])
    for k, (d, n) in agg.items():
        cats[category(k)][0] += d
        cats[category(k)][1] += n
    print("# by category:")
    for k, (d, n) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
        print(f"#   {d / 1e6:8.2f} ms {n:6d} launches  {k}")
    print(f"# {'ms':>8} {'calls':>6} {'avg_us':>9}  kernel")
    for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{d / 1e6:10.3f} {n:6d} {d / n / 1e3:9.1f}  {k[:140]}")


if __name__ == "__main__":
    main()
