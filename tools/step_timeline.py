"""Print the kernel timeline of the second-to-last train step of a rocprofv3 kernel trace.
usage: python tools/step_timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if ('melspec_kernel' in r['Kernel_Name'] or 'melspec_r16_kernel' in r['Kernel_Name'])]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
short = lambda n: n.replace('nafp::', '').replace('void ', '').split('(')[0][:28]
agg = {}
busy = 0.0
for r in rows[s:e]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = short(r['Kernel_Name'])
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    busy += d
    if d >= min_us:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.0f} {d:7.1f} {n:28s} g={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
span = (int(rows[e]['Start_Timestamp']) - t0) / 1e3
print(f'{e - s} launches; span {span:.0f} us; busy {busy:.0f} us')
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:28s} {c:4d} {d:9.1f} us')
