import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mf = [(r['Kernel_Name'].split('(')[0][-40:], float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'scan_mfma' in r['Kernel_Name']]
for name, v, us in mf[-4:]:
    print(f"{name:42s} FETCH_SIZE*2 = {v*1024*2/1e9:8.3f} GB   dur {us:9.1f} us")
