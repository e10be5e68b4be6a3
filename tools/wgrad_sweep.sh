#!/bin/bash
# sweep the workgroup target of the conv3x3 weight gradient (bench.py's instrumented pass, all shapes of the step)
for wgs in 512 1024 2048 4096; do
  DGTD_WGRAD_WGS=$wgs python bench.py --steps 6 --warmup 3 --graph off --no-miou --no-cpu-baseline --all-kernels gpurun_out/kernels_sweep.json > gpurun_out/b_sweep.json 2> gpurun_out/b_sweep.err
  python - "$wgs" <<'PY'
import json, sys
d = json.load(open("gpurun_out/kernels_sweep.json"))
print("wgs =", sys.argv[1], " ".join(f"{k['kernel'].split('[')[1][:-1]}:{k['avg_us']}" for k in d if "conv3x3_wgrad" in k["kernel"]), flush=True)
PY
done
