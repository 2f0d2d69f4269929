"""Per-kernel time inside ONE design iteration (between the starts of the last two k_objective kernels) of a rocprofv3
--kernel-trace database: which kernel is the largest, and its share of the span.   usage: step_shares.py file.db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
obj = [i for i, r in enumerate(rows) if ("k_objective" in r[0][:24])]
# the timed steps come first; the micro-benchmarks of the roofline entries follow the last k_objective
rows = rows[obj[-2]:obj[-1]]
span = rows[-1][2] - rows[0][1]
agg = collections.OrderedDict()
for n, s, e in rows:
    a = agg.setdefault(n.split("(")[0][:72], [0, 0])
    a[0] += 1
    a[1] += e - s
busy = sum(a[1] for a in agg.values())
print("one design iteration under the profiler: span %.2f ms, %d kernels, sum of kernel times %.2f ms" % (span / 1e6, len(rows), busy / 1e6))
print("%-74s %6s %10s %8s %8s" % ("kernel", "calls", "total us", "% span", "% busy"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-74s %6d %10.1f %8.2f %8.2f" % (n, a[0], a[1] / 1e3, 100.0 * a[1] / span, 100.0 * a[1] / busy))
