#!/bin/bash
# both PMC passes for every op of tools/pmc_ops.py (counters in their own runs, kernel trace only)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for op in ${PMC_OPS:-dwconv_k7_32x32x512 dwconv_k7_128x128x128 dwconv_k3_128x128x512 layernorm_8192x512 colsum_8192x2048 conv3x3_96_64x64 conv3x3_Z16_24_128x128 attention_stage1}; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/$op/fetch -- python3 tools/pmc_ops.py $op > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/$op/write -- python3 tools/pmc_ops.py $op > /dev/null 2>&1
  echo "pmc $op done"
done
