#!/bin/bash
# A/B of bench.py variants inside ONE gpurun call (boxes differ by +-10% in host speed): tools/ab_bench.sh "VAR=0" "VAR2=0" ...
# runs the default build and each variant alternately, 2 rounds, and prints ms/step of every run.
set -e
mkdir -p gpurun_out
for round in 1 2; do
  for v in "" "$@"; do
    tag=${v:-default}
    env $v python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag round $round:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
