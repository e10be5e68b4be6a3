#!/bin/bash
# sweep the workgroup target / minimum rows of the batched depthwise weight gradient (bench.py's instrumented pass, all shapes of the step)
for cfg in "1024 8" "2048 8" "4096 8" "2048 4" "8192 4"; do
  set -- $cfg
  DGTD_BWW_BATCHED_WGS=$1 DGTD_BWW_MIN_ROWS=$2 python bench.py --steps 6 --warmup 3 --graph off --no-miou --no-cpu-baseline --all-kernels gpurun_out/kernels_sweep.json > gpurun_out/b_sweep.json 2> gpurun_out/b_sweep.err
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/kernels_sweep.json"))
print("target,min_rows =", sys.argv[1], " ".join(f"{k['kernel'].split('[')[1][:-1]}:{k['avg_us']}" for k in d if "batched" in k["kernel"]), flush=True)
PY
done
