#!/usr/bin/env python
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): true device-side durations of the
hand-written kernels, independent of host launch overhead.   python tools/trace_ops.py <kernel_trace.csv> [substr ...]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:] or ["dwconv", "ln_", "sra_", "attn_delta", "attn_bwd_prep", "diffus", "colsum", "scale_residual"]
agg = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    n = r["Kernel_Name"]
    if not any(p in n for p in pats):
        continue
    short = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", n)[:70]
    key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = agg.setdefault(key, [0, 0, 1 << 60])
    e[0] += d
    e[1] += 1
    e[2] = min(e[2], d)
print(f"{'avg_us':>9} {'min_us':>9} {'calls':>6}  grid(x,y,z) wg vgpr lds scratch  kernel")
for k, (d, n, mn) in agg.items():
    print(f"{d / n / 1e3:9.1f} {mn / 1e3:9.1f} {n:6d}  ({k[1]},{k[2]},{k[3]}) {k[4]} {k[5]} {k[6]} {k[7]}  {k[0]}")
