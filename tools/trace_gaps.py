#!/usr/bin/env python
"""Idle time of the LAST full training step in a rocprofv3 --kernel-trace CSV: the union of all kernel intervals (any stream) against
the step's wall time, and the largest gaps with the kernels on either side.

    python tools/trace_gaps.py <..._kernel_trace.csv> [--top 30]"""
import csv
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
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
    a, b = groups[-2][1] + 1, groups[-1][1] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    gaps, end, last, idle, overlap = [], t0, None, 0, 0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > end:
            idle += s - end
            gaps.append((s - end, last, r["Kernel_Name"], r.get("Queue_Id", "?")))
        else:
            overlap += min(e, end) - s
        if e > end:
            end, last = e, r["Kernel_Name"]
    print(f"# last full step: wall {(t1 - t0) / 1e6:.3f} ms, idle (no kernel on any queue) {idle / 1e6:.3f} ms in {len(gaps)} gaps, "
          f"concurrent kernel time {overlap / 1e6:.3f} ms, {len(step)} launches")
    hist = [0, 0, 0, 0]
    for g in gaps:
        hist[0 if g[0] < 2000 else 1 if g[0] < 10000 else 2 if g[0] < 50000 else 3] += g[0]
    print(f"# idle by gap size: <2us {hist[0] / 1e6:.3f} ms, 2-10us {hist[1] / 1e6:.3f} ms, 10-50us {hist[2] / 1e6:.3f} ms, >50us {hist[3] / 1e6:.3f} ms")
    print("#  gap_us  queue  after -> before")
    for g in sorted(gaps, key=lambda g: -g[0])[:top]:
        print(f"{g[0] / 1e3:9.1f}  {g[3]:>5}  {str(g[1])[:70]}  ->  {g[2][:70]}")
    if "--around" in sys.argv:       # the kernels on either side of the replay boundary (the step's first launches and the previous step's last)
        n = int(sys.argv[sys.argv.index("--around") + 1])
        print(f"# the {n} launches before and after the start of the step (us relative to the step's first launch; start, duration, queue, name)")
        for r in rows[max(0, a - n):a + n]:
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"{(s_ - t0) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f}  {r.get('Queue_Id', '?'):>4}  {r['Kernel_Name'][:90]}")


if __name__ == "__main__":
    main()
