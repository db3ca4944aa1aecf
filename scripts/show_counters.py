"""Per-kernel averages of the counter passes of scripts/prof_counters.sh:  python scripts/show_counters.py TAG [kernel substring]"""
import collections, csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "pyr528")
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(REPO, "gpurun_out", f"{tag}_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(REPO, "gpurun_out", f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        agg[k][c].append(v)
for k in sorted(dur):
    if sub not in k:
        continue
    print(f"{k[:90]}  n={len(dur[k])} avg {sum(dur[k]) / len(dur[k]) / 1e3:.1f} us")
    for c, v in sorted(agg.get(k, {}).items()):
        print(f"    {c:34s} {sum(v) / len(v):16.1f}")
