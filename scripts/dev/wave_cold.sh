#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of scripts/dev/bigkern.py (one / two 8K BGR frames alternating, k = 9, 15, 31: every
# launch reads a tensor the previous launch did not touch) -> per-kernel, per-grid median duration.  $1 = tag
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -- python3 scripts/dev/bigkern.py > gpurun_out/$TAG.log 2>&1 || { echo trace failed; tail -5 gpurun_out/$TAG.log; exit 1; }
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv", recursive=True)[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "sep_" in n: d[(n.split("(anonymous namespace)::")[1].split("(")[0], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    s = sorted(v); print(k, "median %.1f" % s[len(s) // 2], [round(x, 1) for x in v])
PY
