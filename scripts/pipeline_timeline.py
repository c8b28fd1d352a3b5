"""Steady-state picture of the pipelined bench under `rocprofv3 --kernel-trace --output-format csv -d DIR`:
    python scripts/pipeline_timeline.py DIR [n_batches_from_end]
prints every dispatch of the last few batches (start, end, duration in us, gap to the previous end, stream/queue id)
and, per batch, the time no kernel was running."""
import csv
import glob
import sys

d = sys.argv[1]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "prep_queries_kernel" in r[2]]
a = idx[-(nb + 3)]
b = idx[-3]
t0 = rows[a][0]
busy_end = rows[a][0]
idle = 0
for s, e, n, q in rows[a:b]:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} dur={(e - s) / 1e3:8.1f} gap={gap / 1e3:7.1f} q={q}  {n[:70]}")
    busy_end = max(busy_end, e)
print(f"span {(rows[b][0] - t0) / 1e3:.1f} us over {nb} batches, no kernel running for {idle / 1e3:.1f} us")
