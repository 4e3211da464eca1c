#!/usr/bin/env python3
"""Kernel statistics (name, calls, total/avg/min/max ns, %) from a rocprofv3 rocpd SQLite database
-- the same table `rocprofv3 --stats` prints, for builds whose default output is the .db."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
for r in rows:
    nm = r[0].replace("(anonymous namespace)::", "").split("(")[0]
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (nm, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
