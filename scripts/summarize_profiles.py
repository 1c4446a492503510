#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>/ directory written by scripts/collect_profiles.sh into committed summaries:
profiles/<name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<name>_pmc.md (per-kernel HBM-side
traffic from FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes: both in KiB, FETCH_SIZE x2 on
gfx950 for wide coalesced reads), profiles/<name>_bench.json and profiles/pmc_traffic.json (read by bench.py).
usage: python scripts/summarize_profiles.py gpurun_out/prof_r01_c r01"""
import collections, csv, glob, json, os, re, shutil, sys

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "").strip()

shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0], os.path.join(out, f"{name}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, f"{name}_bench.json"))
stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(out, f"{name}_kernel_stats.csv")))}
# the stats CSV averages over EVERY launch, the one-image launches of bench.py's latency section included: the table
# below uses the durations of the full-batch launches only (largest grid of each kernel), from the kernel trace
full_us = {}
tr = glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
if tr:
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        by[short(r["Kernel_Name"])].append((int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for k, v in by.items():
        gmax = max(g for g, _ in v)
        d = sorted(t for g, t in v if g == gmax)
        full_us[k] = (sum(d) / len(d), d[len(d) // 2], len(d))
    with open(os.path.join(out, f"{name}_full_batch_launches.csv"), "w") as f:
        f.write("kernel,launches,mean_us,median_us\n")
        for k, (m, med, cnt) in sorted(full_us.items(), key=lambda kv: -kv[1][0] * kv[1][2]):
            f.write(f"\"{k}\",{cnt},{m:.1f},{med:.1f}\n")
pmc = collections.defaultdict(dict)
for which, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    # full-batch launches only (largest grid of each kernel), as for the durations above: bench.py's latency section
    # launches the same kernels on ONE frame, and a mean over every launch mixes the two (round 2's "0.95 x the
    # algorithmic bytes" for the band kernel and the first r03 table's 0.83 were exactly that mean)
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(src, which, "*", "*counter_collection.csv"))[0])):
        if r["Counter_Name"] == ctr:
            vals[short(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for k, v in vals.items():
        gmax = max(g for g, _ in v)
        full = [x for g, x in v if g == gmax]
        pmc[k][ctr] = sum(full) / len(full)
bench = json.load(open(os.path.join(src, "bench.json")))
B = bench["config"]["batch_per_gpu"]
px = B * 2160 * 3840
alg_by_prefix = {"sep_pow2_band_kernel<3, 5": 6, "sep_pow2_kernel<3, 5": 6, "sep_small_kernel<3, 5>": 6, "conv3_kernel<2, 1>": 2, "conv3_lut_kernel<2>": 3, "sep_pow2_band_gray_kernel<5": 7, "conv3_kernel<0, 3>": 6, "gray_hist_kernel<3, true>": 4,
                 "gray_kernel<3": 4, "lut_kernel<false, true>": 2, "lut_kernel<true, true>": 2, "hist_partial_kernel<": 1}   # the nt template variants share a row


class _Alg:
    def get(self, k):
        for pre, v in alg_by_prefix.items():
            if k.startswith(pre):
                return v
        return None


alg = _Alg()
lines = [f"# {name}: per-kernel duration and HBM-side traffic (batch {B} x 3840x2160)", "",
         "Source: `rocprofv3 --kernel-trace --stats` and two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py`,",
         "collected by scripts/collect_profiles.sh.  Counters are KiB; FETCH_SIZE is doubled (gfx950 reports half the bytes of a",
         "wide coalesced stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as is.  `alg` = algorithmic bytes per launch.", "",
         "| kernel | avg us (full-batch launches) | read MB (2 x FETCH_SIZE) | write MB | traffic MB | alg MB | traffic/alg | alg GB/s | frac of 8 TB/s |",
         "|---|---|---|---|---|---|---|---|---|"]
traffic_json = {}
for k, c in sorted(pmc.items()):
    if k not in stats or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    us = full_us[k][0] if k in full_us else float(stats[k]["AverageNs"]) / 1e3
    rd, wr = 2 * c["FETCH_SIZE"] * 1024 / 1e6, c["WRITE_SIZE"] * 1024 / 1e6
    a = alg.get(k)
    amb = px * a / 1e6 if a else None
    if amb and rd + wr < 0.5 * amb:            # this run only launched the kernel on one frame (the latency section): no batch figures
        a = amb = None
    lines.append(f"| `{k}` | {us:.1f} | {rd:.1f} | {wr:.1f} | {rd + wr:.1f} | " +
                 (f"{amb:.1f} | {(rd + wr) / amb:.3f} | {amb / us * 1e3:.0f} | {amb / us * 1e3 / 8000:.3f} |" if a else "- | - | - | - |"))
    traffic_json[k] = int((rd + wr) * 1e6)
notes = os.path.join(out, f"{name}_pmc_notes.txt")                # hand-written remarks kept next to the generated table
if os.path.exists(notes):
    lines += ["", open(notes).read().rstrip()]
open(os.path.join(out, f"{name}_pmc.md"), "w").write("\n".join(lines) + "\n")
def by_prefix(*prefixes):                                        # (template arguments vary: <3, 5> / <3, 5, true>)
    for pre in prefixes:
        for k, v in traffic_json.items():
            if k.startswith(pre):
                return v
    return None


json.dump({"gaussian5_bgr": by_prefix("sep_pow2_band_kernel<3, 5", "sep_pow2_kernel<3, 5", "sep_small_kernel<3, 5"),
           "gaussian5_bgr+bgr2gray": by_prefix("sep_pow2_band_gray_kernel<5"), "batch_per_gpu": B, "per_kernel_bytes": traffic_json,
           "source": f"profiles/{name}_pmc.md", "recorded_at": f"round {int(name[1:3])}" if name[1:3].isdigit() else name}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
