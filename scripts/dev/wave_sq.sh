#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + one SQ counter pass of waveprof.py cases.  $1 = tag, $2 = k, $3 = cases (mode:tiles,...)
set -u
TAG=$1; K=$2; CASES=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -- python3 scripts/dev/waveprof.py 1 $K $CASES > gpurun_out/$TAG.log 2>&1 || { echo trace failed; tail -3 gpurun_out/$TAG.log; exit 1; }
python3 scripts/dev/waveprof_sum.py gpurun_out/$TAG gpurun_out/$TAG.log
