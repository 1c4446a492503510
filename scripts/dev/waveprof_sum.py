"""dev: durations of the launches waveprof.py made, grouped in its order (kernel_trace.csv + its stdout)."""
import csv, sys, json, glob
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace)) if "sep_wave" in r["Kernel_Name"] or "sep_mfma" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
i = 0
for line in open(sys.argv[2]):
    if not line.startswith("{"): continue
    c = json.loads(line)
    d = durs[i:i + c["launches"]]; i += c["launches"]
    d2 = sorted(d[1:])
    print(f"mode {c['mode']} tiles {c['tiles']:>3s}: median {d2[len(d2)//2]:7.1f} us  min {d2[0]:7.1f}  all {[round(v,1) for v in d]}")
