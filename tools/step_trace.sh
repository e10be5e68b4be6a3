#!/bin/bash
# rocprofv3 kernel trace of bench.py's default command -> per-step kernel summary (tools/trace_summary.py) + the --stats CSV
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/step_trace && mkdir -p gpurun_out/step_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_trace -o run -- python3 bench.py --steps 6 --warmup 3 --no-miou --no-cpu-baseline --profile-steps 0 "$@" > gpurun_out/step_trace/bench.json 2> gpurun_out/step_trace/bench.err
T=$(ls gpurun_out/step_trace/*kernel_trace.csv | head -1)
python3 tools/trace_summary.py $T --top 80 > gpurun_out/step_kernels_new.txt
cp gpurun_out/step_trace/*kernel_stats.csv gpurun_out/step_kernel_stats_new.csv
rm -f $T gpurun_out/step_trace/*.db
head -12 gpurun_out/step_kernels_new.txt
