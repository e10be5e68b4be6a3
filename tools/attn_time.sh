#!/bin/bash
# device durations of the attention kernels at the config-2 stage-1 shape (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/attn_tmp
REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_tmp -- python3 tools/pmc_ops.py attention_stage1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/attn_tmp/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "sra_" in r["Name"] or "attn_delta" in r["Name"] or "dkdv_reduce" in r["Name"]:
        print(f"{r['Name'][:60]:60s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f} us  calls {r['Calls']}")
PY
