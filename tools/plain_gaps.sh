#!/bin/bash
# idle-gap analysis of the plain (N=1) captured step, the counterpart of tools/fa_profile.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/plain; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o run -- python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench.json 2> $O/bench.err
T=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $T --top 25 --around 14 > $O/step_gaps.txt
rm -rf $O/prof
head -70 $O/step_gaps.txt
