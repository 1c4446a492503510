#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of `scripts/cfgbench.py --what 3` (BASELINE configs[2]: 128 x 1080p through the full
# pipeline) -> gpurun_out/<tag>/: per-kernel stats (kernel name, launches, average / min / max us).
set -u
OUT=gpurun_out/${1:-cfg3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/cfgbench.py --what 3 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
grep -v "^W2026\|^E2026" $OUT/trace.log | tail -5
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n.split("(")[0][:60]
    d[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$OUT/kernels.csv", "w") as f:
    f.write("kernel,launches,avg_us,min_us,max_us,total_ms\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.1f,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
PY
cat $OUT/kernels.csv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
