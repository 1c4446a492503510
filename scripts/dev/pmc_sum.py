#!/usr/bin/env python3
"""Average counter values per kernel name from a rocprofv3 --pmc output directory (counter_collection.csv)."""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    if n.startswith("at::") or "elementwise" in n or "distribution" in n: continue
    print(f"{n[:44]:44s} " + " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
