#!/bin/bash
# Runs ON THE GPU BOX: one PMC pass (counters given as arguments) over the large-kernel Gaussian microbench.
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_big/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 scripts/opbench.py --what big > $OUT.log 2>&1 || { echo "pass $TAG failed"; tail -5 $OUT.log; exit 1; }
find $OUT -name "*kernel_trace.csv" -delete
echo done $TAG
