#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of scripts/dev/histeq.py; per kernel and grid: median / average duration.
set -u
OUT=gpurun_out/${1:-heq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/dev/histeq.py > $OUT/trace.log 2>&1 || { echo trace failed; exit 1; }
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "anonymous" in n and "laplacian" not in n and "conv3" not in n:
        d[(n.split("(anonymous namespace)::")[1].split("(")[0], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/trace.txt", "w") as f:
    for k, v in sorted(d.items()):
        s = sorted(v); print(k, "n", len(v), "median %.1f avg %.1f min %.1f" % (s[len(s) // 2], sum(v) / len(v), s[0]), file=f)
PY
grep -v "^W2026\|^E2026" $OUT/trace.log; cat $OUT/trace.txt
