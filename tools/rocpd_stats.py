#!/usr/bin/env python3
"""Export the per-kernel summary of a rocprofv3 rocpd database (.db) to CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/name.csv
"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for r in rows:
        w.writerow([r[0].split('(')[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 2)])
print(open(sys.argv[2]).read())
