#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over waveprof.py (wave kernel, interior or all, warm) -> per-kernel averages.  $1 = tag, $2 = k, $3 = cases
set -u
TAG=$1; K=$2; CASES=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); mkdir -p gpurun_out/$TAG/p$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$TAG/p$i -- python3 scripts/dev/waveprof.py 1 $K $CASES > gpurun_out/$TAG/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$TAG/p$i.log; exit 1; }
  python3 scripts/dev/pmc_grid.py gpurun_out/$TAG/p$i
done
