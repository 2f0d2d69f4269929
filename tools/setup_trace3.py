"""Timeline of the LAST assembly of tools/r04_setup.py (assemblies separated by device synchronisations) from a rocprofv3
--kernel-trace database: every kernel of 10 us or more with stream, start and duration; chains of shorter kernels summarised.
usage: setup_trace3.py file.db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
# the solve after the assemblies starts with k_cheb_first / k_cg...: take the last k_simp before the first CG kernel
cg = next((i for i, r in enumerate(rows) if ("k_cg_" in r[0][:20]) or r[0].startswith("k_cheb_first")), len(rows))
simp = [i for i, r in enumerate(rows[:cg]) if r[0].startswith("k_simp")]
i0 = simp[-1]
seg = rows[i0:cg]
t0 = seg[0][1]
print("set-up span %.3f ms, %d kernels" % ((max(r[2] for r in seg) - t0) / 1e6, len(seg)))
short = collections.defaultdict(lambda: [0, 0.0, None, None])
for n, s, e, st in seg:
    d = (e - s) / 1e3
    if d >= 10.0:
        print("  stream %-2s @%8.1f us  %7.1f us  %s" % (st, (s - t0) / 1e3, d, n.split("(")[0][:60]))
    else:
        a = short[st]
        a[0] += 1; a[1] += d
        a[2] = (s - t0) / 1e3 if a[2] is None else a[2]
        a[3] = (e - t0) / 1e3
for st, a in short.items():
    print("  stream %-2s: %d kernels < 10 us, busy %.0f us, from %.0f to %.0f us" % (st, a[0], a[1], a[2], a[3]))
