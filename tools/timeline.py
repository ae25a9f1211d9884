#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state step from a rocprofv3 kernel trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last occurrence of the first kernel of a step (memset fill of counters precedes k_bin_atoms)
names = [r['Kernel_Name'].split('(')[0] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_sift')]
end = idx[-2]
start = idx[-3] + 1
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:end + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} dur={(e-s)/1e3:6.1f} gap={(s-prev_end)/1e3:6.1f} q={r['Queue_Id']} {r['Kernel_Name'].split('(')[0][:40]}")
    prev_end = max(prev_end, e)
