#!/bin/bash
# rocprofv3 kernel trace of bench.py's default command -> per-step kernel summary (tools/trace_summary.py) + the --stats CSV
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/step_trace && mkdir -p gpurun_out/step_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_trace -o run -- python3 bench.py --steps 6 --warmup 3 --no-miou --no-cpu-baseline --profile-steps 0 "$@" > gpurun_out/step_trace/bench.json 2> gpurun_out/step_trace/bench.err
T=$(ls gpurun_out/step_trace/*kernel_trace.csv | head -1)
python3 tools/trace_summary.py $T --top 80 > gpurun_out/step_kernels_new.txt
python3 - "$T" > gpurun_out/step_long_launches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-2700:]                                   # about one step
for r in tail:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 90:
        print(f"{d:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {r['Kernel_Name'][:110]}")
PY
cp gpurun_out/step_trace/*kernel_stats.csv gpurun_out/step_kernel_stats_new.csv
rm -f $T gpurun_out/step_trace/*.db
head -12 gpurun_out/step_kernels_new.txt
