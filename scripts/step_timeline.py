"""Print the kernel timeline of the last search step in a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'final_topk' in r['Kernel_Name']]
j = idx[-1]; k = idx[-2]
seg = rows[k + 1:j + 1]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} dur={(e-s)/1e3:9.1f} us  {r['Kernel_Name'][:64]}")
