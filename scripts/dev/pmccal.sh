#!/bin/bash
# Runs ON THE GPU BOX: TCC read counters of the calibration kernels (scripts/dev/pmccal.hip) and of the bench's kernels.  $1 = tag
set -u
TAG=${1:-pmccal}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/cal$i -- scripts/dev/_build/pmccal > $OUT/cal$i.log 2>&1 || { echo "cal pass $i failed"; tail -3 $OUT/cal$i.log; exit 1; }
  echo "== calibration: $C"; python3 scripts/dev/pmc_grid.py $OUT/cal$i | tee $OUT/cal$i.txt
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/bench$i -- python3 bench.py --cpu-sample 0 --steps 2 --warmup 1 > $OUT/bench$i.log 2>&1 || { echo "bench pass $i failed"; tail -3 $OUT/bench$i.log; exit 1; }
  echo "== bench.py: $C"; python3 scripts/dev/pmc_grid.py $OUT/bench$i | tee $OUT/bench$i.txt
  find $OUT -name "*.csv" -size +2M -delete
done
