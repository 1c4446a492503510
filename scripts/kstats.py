#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly. usage: python scripts/kstats.py <dir or csv>"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "*", "*kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    print("%-46s calls=%5s avg=%9.1fus min=%9.1f max=%9.1f" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
