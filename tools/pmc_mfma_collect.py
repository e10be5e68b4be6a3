#!/usr/bin/env python
"""Summarise the MFMA PMC passes of tools/pmc_mfma.sh: per attention kernel, counters averaged per launch and the derived MFMA
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x clock x 256 CUs x 4 SIMDs) (busy cycles count per SIMD, guide constants
table) beside the flop-based figure."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for sub in ("a", "b"):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*|void ", "", r["Kernel_Name"])[:64]
            if not any(k in name for k in ("sra_", "attn_delta", "gemm_tn")):
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*|void ", "", r["Kernel_Name"])[:64]
            if any(k in name for k in ("sra_", "attn_delta", "gemm_tn")):
                dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# attention kernels at B=8, N=16384, Nkv=256, h=1 and the own-GEMM kernels of one convnext_Block pointwise half at [8192, 512] (bf16):\n# PMC counters per launch (mean), rocprofv3 --pmc, ROCm 7.2, gfx950")
for name, ctrs in agg.items():
    d = sum(dur[name]) / max(len(dur[name]), 1)
    print(f"\n{name}   launches {len(dur[name]) // 2}   avg duration {d / 1e3:.1f} us (under the profiler)")
    for c, v in sorted(ctrs.items()):
        print(f"   {c:34s} {sum(v) / len(v):16.0f}")
    mf = ctrs.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if mf and d > 0:
        m = sum(mf) / len(mf)
        for ghz in (2.4, 2.0):
            print(f"   MFMA busy / (duration x {ghz} GHz x 1024 SIMDs)    {m / (d * ghz * 1024):.3f}")
