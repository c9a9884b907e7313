"""Per-kernel mean of PMC counters from rocprofv3 rocpd databases.
usage: python scripts/rocpd_pmc.py db1 [db2 ...] [--match substr]"""
import sqlite3
import sys

match = None
dbs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        dbs.append(a)
for path in dbs:
    db = sqlite3.connect(path)
    cur = db.cursor()
    try:
        rows = cur.execute(
            "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name").fetchall()
    except sqlite3.OperationalError as e:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print(path, "schema:", cols, e)
        continue
    for k, c, v, n in rows:
        if match and match not in k:
            continue
        print(f"{k[:60]:60s} {c:24s} {v:18.1f}  (n={n})")
