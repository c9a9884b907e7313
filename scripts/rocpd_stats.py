"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total, mean, share.
usage: python scripts/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
total = sum(r[2] for r in rows) or 1
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")]
for n, c, t, a, mn, mx in rows:
    out.append((n, c, t, round(a, 1), mn, mx, round(100.0 * t / total, 2)))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        csv.writer(f).writerows(out)
for r in out[:45]:
    print(f"{str(r[0])[:100]:100s} {str(r[1]):>6s} {str(r[2]):>14s} {str(r[3]):>12s} {str(r[6]):>7s}")
