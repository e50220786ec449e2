#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max) of a rocprofv3 --kernel-trace rocpd database."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-78s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows:
    print("%-78s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
