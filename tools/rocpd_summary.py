#!/usr/bin/env python3
"""Kernel summary (calls, total / average / min / max duration) from a rocprofv3 rocpd SQLite database.
Usage: rocpd_summary.py results.db > profiles/NAME_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
