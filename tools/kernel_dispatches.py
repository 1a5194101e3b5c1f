#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace run (rocpd sqlite output): the launches of a
program that mixes launch lengths (bench.py at the driver's --steps 20: 125-step spin-up launches, a 5-step warm-up launch,
the timed 20-step launch) cannot be read off the per-kernel average.
usage: kernel_dispatches.py <results.db> <name substring> <out.txt> [header comment ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)).fetchall()
with open(sys.argv[3], "w") as f:
    for c in sys.argv[4:]:
        f.write("# " + c + "\n")
    f.write("# dispatch, start_us (from the first), duration_us\n")
    t0 = rows[0][1] if rows else 0
    for i, (name, start, dur) in enumerate(rows):
        f.write("%d, %.1f, %.2f\n" % (i, (start - t0) / 1e3, dur / 1e3))
print(len(rows), "dispatches")
