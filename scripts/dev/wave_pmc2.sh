#!/bin/bash
# Runs ON THE GPU BOX: memory-side PMC passes.  $1 = tag, then the python command (script + args) to profile
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); mkdir -p gpurun_out/$TAG/m$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d gpurun_out/$TAG/m$i -- python3 "$@" > gpurun_out/$TAG/m$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$TAG/m$i.log; exit 1; }
  python3 scripts/dev/pmc_grid.py gpurun_out/$TAG/m$i | grep "48, 2"
  find gpurun_out/$TAG/m$i -name "*.csv" -size +4M -delete
done
