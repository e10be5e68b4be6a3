#!/bin/bash
# the round's committed evidence: default bench line (+ rocprofv3 kernel stats of the same command), per-step kernel summary, and the
# bench lines of configs 3 / 4 / 5 and of the forced-all-reduce (split graph) rehearsal.  Outputs under gpurun_out/final/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python3 bench.py --all-kernels $O/kernels_all.json > $O/bench.json 2> $O/bench.err
echo "bench: $(python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --no-miou --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
T=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 tools/trace_summary.py $T --top 80 > $O/step_kernels.txt
cp $O/prof/*kernel_stats.csv $O/bench_rocprof_kernel_stats.csv
rm -rf $O/prof
head -11 $O/step_kernels.txt
python3 bench.py --backbone pvt_v2_b3 --no-miou --no-cpu-baseline > $O/bench_config3.json 2> $O/c3.err && echo "config3 done"
python3 bench.py --mode predict --size 1024 --batch 4 --dtype f32 --no-miou --no-cpu-baseline > $O/bench_config4.json 2> $O/c4.err && echo "config4 done"
python3 bench.py --dtype f16 --batch 16 --no-miou --no-cpu-baseline > $O/bench_config5.json 2> $O/c5.err && echo "config5 done"
DGTD_FORCE_ALLREDUCE=1 python3 bench.py --no-miou --no-cpu-baseline > $O/bench_forced_allreduce.json 2> $O/fa.err && echo "forced allreduce done"
for f in bench_config3 bench_config4 bench_config5 bench_forced_allreduce; do python3 -c "import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f', d['value'], d['ms_per_step'], d['config'].get('hip_graph'))"; done
