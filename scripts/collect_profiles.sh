#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats of the default bench command and two PMC passes
# (FETCH_SIZE, WRITE_SIZE in runs of their own, as the MI355X guide prescribes).  Output: gpurun_out/<tag>/
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --ops-table > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-sample 0 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --cpu-sample 0 --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --cpu-sample 0 --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
# keep the merged artefacts small: kernel_trace CSVs of PMC runs can be large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT $OUT/*/* | head -40
cat $OUT/bench.json
