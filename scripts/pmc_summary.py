"""Per-kernel summary of a rocprofv3 counter pass:  python scripts/pmc_summary.py DIR [kernel_substring]
Reads every *counter_collection.csv under DIR; prints, per kernel name, dispatches and per counter the sum and the
mean per dispatch.  FETCH_SIZE / WRITE_SIZE come in KiB: the bytes column applies x1024, and for FETCH_SIZE the x2 of
gfx950's half-count on wide coalesced streams (MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = len(disp[k])
    print(f"{k[:110]}  dispatches={n}")
    for c, v in sorted(acc[k].items()):
        extra = ""
        if c == "FETCH_SIZE":
            extra = f"   -> {v * 1024 * 2 / 1e9:.3f} GB read in all ({v * 1024 * 2 / n / 1e9:.4f} GB per dispatch; KiB x 1024 x 2)"
        if c == "WRITE_SIZE":
            extra = f"   -> {v * 1024 / 1e9:.3f} GB written in all"
        print(f"    {c:40s} sum={v:.6g}  per_dispatch={v / n:.6g}{extra}")
