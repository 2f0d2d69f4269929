"""GPU idle time between kernels from a rocprofv3 --kernel-trace database: usage gaps.py file.db
One design iteration = the span between the starts of the last two k_objective kernels.  Prints busy/idle and the
kernels that precede the idle gaps."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
obj = [i for i, r in enumerate(rows) if ("k_objective" in r[0][:24])]
rows = rows[obj[-2]:obj[-1]]
t0, t1 = rows[0][1], rows[-1][2]
busy = 0; cur_e = rows[0][1]; prev = ""; gaps = collections.defaultdict(lambda: [0, 0]); hist = collections.Counter()
for name, s, e in rows:
    if s > cur_e:
        g = s - cur_e
        gaps[prev][0] += 1; gaps[prev][1] += g
        hist[min(int(g / 1000), 50)] += g
    busy_s = max(s, cur_e)
    if e > busy_s: busy += e - busy_s
    if e > cur_e: cur_e = e; prev = name.split("(")[0][:60]
span = t1 - t0
print("one design iteration: span %.2f ms, %d kernels, busy %.2f ms, idle %.2f ms (%.1f %%)" % (span / 1e6, len(rows), busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
print("idle by gap length (us bucket: total us):", {k: round(v / 1e3) for k, v in sorted(hist.items())})
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%8.1f us total idle after %-60s x%d  avg %.2f us" % (g / 1e3, k, c, g / 1e3 / c))
