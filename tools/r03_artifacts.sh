#!/bin/bash
# Round-3 evidence, one GPU call: default bench line (+ every kernel record), rocprofv3 kernel-trace stats of the same command, per-step
# kernel summary, PMC traffic passes for the roofline shapes, MFMA PMC passes (attention stage 1, own GEMM), the other BASELINE configs,
# the forced-all-reduce rehearsal in both captured forms, and the slowest GPU tests.  Outputs under gpurun_out/final/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python3 bench.py --all-kernels $O/kernels_all.json > $O/bench.json 2> $O/bench.err
echo "bench: $(python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
T=$(ls $O/prof/*kernel_trace.csv | head -1)
python3 tools/trace_summary.py $T --top 90 > $O/step_kernels.txt
cp $O/prof/*kernel_stats.csv $O/bench_rocprof_kernel_stats.csv
rm -rf $O/prof
head -11 $O/step_kernels.txt
PMC_OPS="mlp_residual_8192x512 dwconv_k7_32x32x512 dwconv_k7_128x128x128 dwconv_k3_128x128x512 layernorm_8192x512 attention_stage1" bash tools/pmc_run.sh > $O/pmc_run.log 2>&1 || echo "pmc_run failed"
python3 tools/pmc_collect.py gpurun_out/pmc $O/pmc_by_bench_key.json > $O/pmc_traffic.json 2> $O/pmc_collect.err || echo "pmc_collect failed"
bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1 || echo "pmc_mfma failed"
cp gpurun_out/pmc_mfma/summary.txt $O/pmc_mfma.txt 2>/dev/null || true
rm -rf gpurun_out/pmc gpurun_out/pmc_mfma/a gpurun_out/pmc_mfma/b
python3 bench.py --backbone pvt_v2_b3 --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_config3.json 2> $O/c3.err && echo "config3 done"
python3 bench.py --mode predict --size 1024 --batch 4 --dtype f32 --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_config4.json 2> $O/c4.err && echo "config4 done"
python3 bench.py --dtype f16 --batch 16 --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_config5.json 2> $O/c5.err && echo "config5 done"
DGTD_FORCE_ALLREDUCE=1 python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_forced_allreduce_fused.json 2> $O/fa1.err && echo "forced allreduce (fused) done"
DGTD_FORCE_ALLREDUCE=1 DGTD_GRAPH_COMM=split python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 > $O/bench_forced_allreduce_split.json 2> $O/fa2.err && echo "forced allreduce (split) done"
for f in bench_config3 bench_config4 bench_config5 bench_forced_allreduce_fused bench_forced_allreduce_split; do python3 -c "import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f', d['value'], d['ms_per_step'], d['config'].get('graph_mode'), d['config'].get('final_loss'))"; done
