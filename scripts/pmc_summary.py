"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the dispatch with the largest grid*duration."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "scan_mfma"
best = {}
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = r["Dispatch_Id"]
    best.setdefault(key, {"dur": dur, "name": r["Kernel_Name"][:60], "c": {}})
    best[key]["c"][r["Counter_Name"]] = float(r["Counter_Value"])
top = sorted(best.values(), key=lambda x: -x["dur"])[:1]
for t in top:
    print(t["name"], "dur_us=%.1f" % (t["dur"] / 1e3))
    for k, v in sorted(t["c"].items()):
        print(f"   {k:36s} {v:16.0f}")
