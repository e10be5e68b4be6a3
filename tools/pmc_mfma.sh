#!/bin/bash
# MFMA-utilisation PMC pass for the attention kernels at the config-2 stage-1 shape (counters in their own run, kernel trace only):
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE-equivalent busy cycles x CUs x 4 SIMDs)   (gfx94x formula; see the guide)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_mfma
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CY|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU" | head -40 > gpurun_out/pmc_mfma/counters_available.txt || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/a -- python3 tools/pmc_ops.py attention_stage1 > /dev/null 2>&1 || echo "pass a failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/b -- python3 tools/pmc_ops.py attention_stage1 > /dev/null 2>&1 || echo "pass b failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/a -- python3 tools/pmc_ops.py mlp_residual_8192x512 > /dev/null 2>&1 || echo "pass a (gemm) failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_mfma/b -- python3 tools/pmc_ops.py mlp_residual_8192x512 > /dev/null 2>&1 || echo "pass b (gemm) failed"
python3 tools/pmc_mfma_collect.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma/summary.txt
cat gpurun_out/pmc_mfma/summary.txt
