#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of scripts/dev/onestep.py; prints, per phase, each kernel's median duration and the
# median gap to the previous kernel's end.
set -u
OUT=gpurun_out/${1:-onestep}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/dev/onestep.py > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
grep -v "^W2026\|^E2026" $OUT/trace.log
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "fill" in n.lower() or "FillFunctor" in n:
        cur = []; steps.append(cur); continue
    if cur is not None and "anonymous" in n:
        cur.append((n.split("(anonymous namespace)::")[1].split("(")[0][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
steps = [s for s in steps if len(s) >= 4]
# three phases of 30 steps each (the 5 warm-up steps of each phase have no marker)
for name, chunk in (("eager", steps[:30]), ("graph", steps[30:60]), ("forked", steps[60:90])):
    if not chunk: continue
    print(name, "steps", len(chunk))
    k = len(chunk[0])
    for j in range(k):
        d = sorted((s[j][2] - s[j][1]) / 1e3 for s in chunk if len(s) == k)
        gp = sorted((s[j][1] - s[j - 1][2]) / 1e3 for s in chunk if len(s) == k) if j else [0]
        print("   %-42s dur %.1f  gap before %.1f" % (chunk[0][j][0], d[len(d) // 2], gp[len(gp) // 2]))
    tot = sorted((max(x[2] for x in s) - s[0][1]) / 1e3 for s in chunk if len(s) == k)
    print("   first start -> last end %.1f" % tot[len(tot) // 2])
PY
