#!/bin/bash
# Runs ON THE GPU BOX: the r03 record of the 9..31-tap kernels on 8K BGR frames (BASELINE configs[3]).
#   1. rocprofv3 --kernel-trace of scripts/dev/bigkern.py (one / two frames alternating: every launch follows a launch on other tensors)
#   2. FETCH_SIZE and WRITE_SIZE passes of the same script
#   3. kernel trace of scripts/dev/coldwhy.py (k = 31, 15): the same launch after nothing / after kernels that flush the caches
# Output: gpurun_out/$1/{trace,fetch,write,why31,why15}*
set -u
TAG=${1:-r03_mfma}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/dev/bigkern.py > $OUT/trace.log 2>&1 || { echo trace failed; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 scripts/dev/bigkern.py > $OUT/$C.log 2>&1 || { echo "$C failed"; exit 1; }
  python3 scripts/dev/pmc_grid.py $OUT/$C > $OUT/$C.txt
done
for K in 31 15; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/why$K -- python3 scripts/dev/coldwhy.py $K > $OUT/why$K.log 2>&1 || { echo why failed; exit 1; }
  python3 scripts/dev/coldwhy_sum.py $OUT/why$K $OUT/why$K.log > $OUT/why$K.txt
done
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "sep_" in n: d[(n.split("(anonymous namespace)::")[1].split("(")[0], r["Grid_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/trace.txt", "w") as f:
    for k, v in sorted(d.items()):
        s = sorted(v); print(k, "median %.1f avg %.1f" % (s[len(s) // 2], sum(v) / len(v)), [round(x, 1) for x in v], file=f)
PY
cat $OUT/trace.txt $OUT/FETCH_SIZE.txt $OUT/WRITE_SIZE.txt $OUT/why31.txt $OUT/why15.txt
find $OUT -name "*.csv" -size +2M -delete
