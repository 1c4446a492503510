#!/usr/bin/env python3
"""Average counter values per (kernel, grid size) from a rocprofv3 --pmc output directory (counter_collection.csv)."""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    if n.startswith("at::") or "elementwise" in n or "distribution" in n: continue
    acc[(n, r.get("Grid_Size", r.get("Grid_Size_X", "")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, g), cs in sorted(acc.items()):
    print(f"{n[:40]:40s} grid {g:>8s} " + " ".join(f"{c}={sum(v) / len(v):.5g}" for c, v in sorted(cs.items())))
