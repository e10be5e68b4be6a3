#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for op in dwconv_k7_32x32x512 dwconv_k7_128x128x128 dwconv_k3_128x128x512; do
for t in 256 512 1024 2048 4096; do
  export DGTD_BWW_TARGET_K7=$t DGTD_BWW_TARGET_K3=$t
  rm -rf gpurun_out/sweep_tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_tmp -- python3 tools/pmc_ops.py $op > /dev/null 2>&1
  python3 - "$op" "$t" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/sweep_tmp/*/*kernel_stats.csv")[0]
out = []
for r in csv.DictReader(open(f)):
    if "dwconv_bwd_weight_kernel" in r["Name"] or "dwconv_bww_reduce" in r["Name"]:
        out.append(("bww" if "bwd_weight" in r["Name"] else "reduce") + f" {float(r['AverageNs'])/1e3:7.1f} us")
print(sys.argv[1], "target", sys.argv[2], " | ".join(sorted(out)))
PY
done; done
