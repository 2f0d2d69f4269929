#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max ns, % of GPU time) of a rocprofv3
--kernel-trace --stats result database (rocpd SQLite) -> CSV on stdout."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, (end - start) from kernels").fetchall()
agg = {}
for name, d in rows:
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
