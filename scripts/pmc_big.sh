#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over the large-kernel Gaussian microbench (one pass per counter group).
set -u
OUT=gpurun_out/pmc_big
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/opbench.py --what big > $OUT/fetch.log 2>&1 || { echo "fetch failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/opbench.py --what big > $OUT/write.log 2>&1 || { echo "write failed"; tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 scripts/opbench.py --what big > $OUT/sq.log 2>&1 || { echo "sq failed"; tail -5 $OUT/sq.log; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
echo done
