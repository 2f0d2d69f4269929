"""One CG iteration (between the last two k_cg_update_xr of the timed steps) of a rocprofv3 --kernel-trace database, kernel by
kernel: start offset, duration, idle gap before, grid.   usage: iter_timeline.py file.db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
obj = [i for i, r in enumerate(rows) if ("k_objective" in r[0][:24])]
if len(obj) >= 2:
    rows = rows[obj[-2]:obj[-1]]
xr = [i for i, r in enumerate(rows) if "k_cg_update_xr" in r[0][:40]]
seg = rows[xr[-3]:xr[-2] + 1]
t0 = seg[0][1]
prev_end = seg[0][1]
tot_busy = 0
print("one CG iteration: %.1f us, %d kernels" % ((seg[-1][1] - t0) / 1e3, len(seg) - 1))
for n, s, e, gx, wx in seg[:-1]:
    print("%8.1f  dur %6.1f  gap %5.1f  wgs %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, gx // max(wx, 1), n.split("(")[0][:70]))
    prev_end = max(prev_end, e)
    tot_busy += e - s
print("busy %.1f us" % (tot_busy / 1e3))
