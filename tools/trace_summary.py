#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV for ONE steady-state training step.

    python tools/trace_summary.py gpurun_out/prof/.../*_kernel_trace.csv [--top 40] > profiles/rNN_step_kernels.txt

Steps are delimited by the fused-AdamW kernels (multi_tensor_apply); the LAST full step is reported so
MIOpen's first-step solver search and allocator warm-up are excluded."""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
    groups, prev = [], None
    for i in opt:
        if prev is None or i - prev > 50:
            groups.append([i, i])
        else:
            groups[-1][1] = i
        prev = i
    a, b = groups[-2][1] + 1, groups[-1][1] + 1
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b - 1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    busy = 0
    for r in rows[a:b]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += d
        agg[r["Kernel_Name"]][1] += 1
        busy += d
    print(f"# last full step: wall {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, {b - a} kernel launches")
    print(f"# {'ms':>8} {'calls':>6} {'avg_us':>9}  kernel")
    for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{d / 1e6:10.3f} {n:6d} {d / n / 1e3:9.1f}  {k[:140]}")


if __name__ == "__main__":
    main()
