"""Timeline of the last batch of a run under `rocprofv3 --kernel-trace --output-format csv -d DIR`:
    python scripts/step_timeline.py DIR [first_kernel_substring]
prints start / end / duration (us) of every dispatch from the last launch of the batch's first kernel
(default: prep_queries_kernel) to the end of the trace."""
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prep_queries_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if first in r[2]]
if not idx:
    sys.exit("no dispatch of " + first)
# the last COMPLETE batch: from the second-to-last first-kernel to the last one
a, b = (idx[-2], idx[-1]) if len(idx) > 1 else (idx[-1], len(rows))
t0 = rows[a][0]
for s, e, n in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} dur={(e - s) / 1e3:9.1f} us  {n[:90]}")
