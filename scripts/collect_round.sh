#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the round's records in one call -- scripts/collect_profiles.sh (ops-table bench line,
# kernel trace + stats of the default bench command, FETCH_SIZE / WRITE_SIZE passes), a kernel trace of the ops-table run
# (scripts/summarize_ops_trace.py reads it), scripts/cfgbench.py (configs[2..4] on one GPU) and bench.py --workload rowpart.
# usage: bash scripts/collect_round.sh r03_final
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT/ops
bash scripts/collect_profiles.sh $TAG > $OUT/collect.log 2>&1 || { tail -5 $OUT/collect.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/ops/trace -- python3 bench.py --cpu-sample 0 --ops-table > $OUT/ops/trace.log 2>&1 || { echo "ops trace failed"; tail -5 $OUT/ops/trace.log; exit 1; }
find $OUT/ops -name "*kernel_trace.csv" -size +16M -delete
timeout -k 10 300 python3 scripts/cfgbench.py > $OUT/cfgbench.log 2>&1 || { echo "cfgbench failed"; tail -5 $OUT/cfgbench.log; exit 1; }
timeout -k 10 200 python3 bench.py --workload rowpart --cpu-sample 0 > $OUT/rowpart.json 2> $OUT/rowpart.err || { echo "rowpart failed"; tail -5 $OUT/rowpart.err; exit 1; }
tail -3 $OUT/cfgbench.log; cat $OUT/rowpart.json
