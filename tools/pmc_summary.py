#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite output): per-kernel mean of every counter.
usage: pmc_summary.py <results.db> [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_train_steps"
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                  "where kernel_name like ? group by kernel_name, counter_name", ("%" + pat + "%",)).fetchall()
for name, ctr, n, val, dur in rows:
    print(f"{name[:60]:60s} {ctr:28s} n={n:4d} mean={val:16.1f} mean_dur_us={dur / 1e3:10.1f}")
