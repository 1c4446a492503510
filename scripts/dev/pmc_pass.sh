#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 PMC pass (counters as arguments after the tag and the python script) -> gpurun_out/pmc/<tag>
set -u
TAG=$1; SCRIPT=$2; shift 2
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $SCRIPT > $OUT.log 2>&1 || { echo "pass $TAG failed"; tail -5 $OUT.log; exit 1; }
python3 scripts/dev/pmc_sum.py $OUT
