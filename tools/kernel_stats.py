#!/usr/bin/env python3
"""Export the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as CSV.
usage: kernel_stats.py <results.db> <out.csv> [header comment ...]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    for c in sys.argv[3:]:
        f.write("# " + c + "\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name[:110], calls, round(tot, 3), round(avg, 3), round(pct, 4)])
