#!/bin/bash
# kernel trace of the forced-all-reduce rehearsal (world-1 RCCL, captured "fused" mode): where the N>1 step spends more than the plain one
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fa; rm -rf $O; mkdir -p $O
export DGTD_FORCE_ALLREDUCE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench.json 2> $O/bench.err
T=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 tools/trace_summary.py $T --top 120 > $O/step_kernels.txt
python3 tools/trace_gaps.py $T --top 40 > $O/step_gaps.txt
rm -rf $O/prof
head -12 $O/step_kernels.txt; head -45 $O/step_gaps.txt
