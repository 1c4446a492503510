#!/usr/bin/env python3
"""profiles/<round>_ops_table_kernels.csv from a `rocprofv3 --kernel-trace` run of `bench.py --cpu-sample 0 --ops-table`:
per kernel, the launches with its largest grid (the full-batch launches of the step and of the ops table) -- count and
median / min / max duration.

    python scripts/summarize_ops_trace.py gpurun_out/<dir with trace/> r02
"""
import csv, glob, os, re, statistics, sys

src, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv"))[0]
by = {}
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    by.setdefault(r["Kernel_Name"], []).append((grid, d))
out = []
for name, v in by.items():
    gmax = max(g for g, _ in v)
    ds = sorted(d for g, d in v if g == gmax)
    short = re.sub(r"\((?!anonymous).*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    out.append((short, len(ds), round(statistics.median(ds), 1), round(ds[0], 1), round(ds[-1], 1)))
out.sort(key=lambda x: -x[2])
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_ops_table_kernels.csv")
with open(dst, "w") as g:
    g.write("# rocprofv3 --kernel-trace of `python3 bench.py --cpu-sample 0 --ops-table` (24 x 4K BGR; gray operators on the 24 x 4K gray batch):\n")
    g.write("# per kernel, the launches with its largest grid (= full-batch launches of the step and of the ops table): count, median / min / max duration in us\n")
    g.write("kernel,full_batch_launches,median_us,min_us,max_us\n")
    for o in out:
        g.write(f"\"{o[0]}\",{o[1]},{o[2]},{o[3]},{o[4]}\n")
print("wrote", os.path.relpath(dst))
