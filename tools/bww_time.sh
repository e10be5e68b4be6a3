#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for op in dwconv_k7_32x32x512 dwconv_k7_128x128x128 dwconv_k3_128x128x512; do
  rm -rf gpurun_out/sweep_tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_tmp -- python3 tools/pmc_ops.py $op > /dev/null 2>&1
  python3 - "$op" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/sweep_tmp/*/*kernel_stats.csv")[0]
out = []
for r in csv.DictReader(open(f)):
    if "dwconv" in r["Name"]:
        out.append(f"{r['Name'][20:58]} {float(r['AverageNs'])/1e3:7.1f} us")
print(sys.argv[1], " | ".join(sorted(out)))
PY
done
