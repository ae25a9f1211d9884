#!/usr/bin/env python3
"""Kernel timeline of the LAST fresh-structure step of tools/fresh_probe.py from a rocprofv3 kernel trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for extra in sys.argv[2:]:       # (memory-copy trace of the same run: copies appear as rows of their own)
    for r in csv.DictReader(open(extra)):
        rows.append({'Kernel_Name': 'COPY_' + r.get('Direction', r.get('Name', '?')), 'Start_Timestamp': r['Start_Timestamp'], 'End_Timestamp': r['End_Timestamp']})
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('void ', '') for r in rows]
first = [i for i, n in enumerate(names) if n.startswith('k_validate_blob')]
start = first[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r, n in zip(rows[start:], names[start:]):
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} dur={(e-s)/1e3:6.1f} gap={max(s-prev_end, -999000)/1e3:6.1f} {n[:50]}")
    prev_end = max(prev_end, e)
