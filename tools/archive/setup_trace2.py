"""Per-stream timeline of the LAST set-up (k_matfree_diag .. end of trace or next k_cheb_first) of a rocprofv3 database
made with --kernel-trace [--hip-trace]; with the HIP API trace also the host-side span of the calls that enqueued it.
usage: setup_trace2.py file.db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if r[0].startswith("void k_matfree_diag")]
i0 = starts[-1]
firsts = [i for i, r in enumerate(rows) if r[0].startswith("k_cheb_first") and i > i0]
i1 = firsts[0] if firsts else len(rows)
# the early pass (Galerkin chain to the coarsest level + factorisation) starts before k_matfree_diag: go back to the first kernel after a gap > 200 us
j = i0
while j > 0 and rows[j][1] - rows[j - 1][2] < 200000 and i0 - j < 400:
    j -= 1
seg = rows[j:i1]
t0 = seg[0][1]
print("set-up span %.3f ms, %d kernels (from the first kernel after an idle gap to the last one)" % ((max(r[2] for r in seg) - t0) / 1e6, len(seg)))
by = collections.OrderedDict()
for r in seg:
    by.setdefault(r[3], []).append(r)
for k, v in by.items():
    names = collections.Counter(x[0].split("(")[0][:40] for x in v)
    print("stream %s: %4d kernels, from %.3f to %.3f ms, busy %.3f ms | %s" % (k, len(v), (v[0][1] - t0) / 1e6, (max(x[2] for x in v) - t0) / 1e6,
          sum(x[2] - x[1] for x in v) / 1e6, ", ".join("%s x%d" % kv for kv in names.most_common(5))))
big = sorted(seg, key=lambda r: r[1] - r[2])[:8]
print("longest kernels:", ", ".join("%s %.0f us @%.3f" % (r[0].split("(")[0][:28], (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e6) for r in big))
api = [t for t in tabs if "region" in t.lower() or "api" in t.lower()]
print("tables:", tabs)
for t in ("regions", "regions_and_samples"):
    if t in tabs:
        c = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
        print(t, c)
        try:
            rr = con.execute("select name, start, end from %s where start >= ? and start <= ? order by start" % t, (t0 - 3000000, max(r[2] for r in seg))).fetchall()
            if rr:
                print("host API calls in the window: %d, from %.3f to %.3f ms" % (len(rr), (rr[0][1] - t0) / 1e6, (rr[-1][2] - t0) / 1e6))
                cnt = collections.Counter(x[0] for x in rr)
                tot = collections.Counter()
                for x in rr: tot[x[0]] += x[2] - x[1]
                print(", ".join("%s x%d %.0f us" % (k, v, tot[k] / 1e3) for k, v in cnt.most_common(10)))
                gl = [x for x in rr if "GraphLaunch" in x[0]]
                for x in gl: print("   %s at %.3f .. %.3f ms" % (x[0], (x[1] - t0) / 1e6, (x[2] - t0) / 1e6))
        except Exception as e:
            print("query failed", e)
        break
