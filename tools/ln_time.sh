#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sweep_tmp
REPS=30 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_tmp -- python3 tools/pmc_ops.py layernorm_8192x512 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sweep_tmp/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "ln_" in r["Name"]:
        print(f"{r['Name'][:60]:60s} {float(r['AverageNs'])/1e3:7.1f} us  x{r['Calls']}")
PY
