#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another) per kernel+grid.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide coalesced reads by exactly 2x -> doubled here;
units are KiB.   python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import collections
import csv
import re
import sys


def load(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", r["Kernel_Name"])[:64]
        key = (name, r["Grid_Size"])
        e = agg.setdefault(key, [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1
    return agg


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
print(f"{'fetch_MB(x2)':>12} {'write_MB':>9} {'launches':>8}  grid  kernel")
for key, (f, n) in fetch.items():
    if not any(k in key[0] for k in ("dwconv", "ln_", "sra_", "diffus", "colsum", "scale_residual", "loss_")):
        continue
    w = write.get(key, [0.0, 1])
    print(f"{2 * f / n * 1024 / 1e6:12.2f} {w[0] / max(w[1], 1) * 1024 / 1e6:9.2f} {n:8d}  {key[1]:>9}  {key[0]}")
