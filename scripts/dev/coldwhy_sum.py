import csv, sys, json, glob
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace)) if ("sep_wave" in r["Kernel_Name"] or "sep_mfma" in r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grids = sorted(set(int(r["Grid_Size_X"]) for r in rows))
rows = [r for r in rows if int(r["Grid_Size_X"]) == grids[0]]                     # the one-frame launches
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
i = 0
for line in open(sys.argv[2]):
    if not line.startswith("{"): continue
    c = json.loads(line); d = durs[i:i + c["launches"]]; i += c["launches"]
    s = sorted(d[1:])
    print(f"after {c['case']:>14s}: median {s[len(s)//2]:6.1f} us  {[round(v, 1) for v in d]}")
