"""Timeline of the set-up part of ONE design iteration (k_simp .. first k_cheb_first) of a rocprofv3 --kernel-trace
database, per stream: where the spectra chains of the levels sit and which one the solve waits for.
usage: setup_trace.py file.db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = con.execute("select name, start, end%s from kernels order by start" % (", " + key if key else "")).fetchall()
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_simp")] or [i for i, r in enumerate(rows) if r[0].startswith("void k_matfree_diag")]
firsts = [i for i, r in enumerate(rows) if r[0].startswith("k_cheb_first")]
i0 = [i for i in starts if i < firsts[-1]][-1]
i1 = next(i for i in firsts if i > i0)
seg = rows[i0:i1]
t0 = seg[0][1]
print("columns: %s; grouping by %s" % (",".join(cols), key))
print("set-up span %.3f ms, %d kernels" % ((rows[i1][1] - t0) / 1e6, len(seg)))
by = collections.OrderedDict()
for r in seg:
    by.setdefault(r[3] if key else 0, []).append(r)
for k, v in by.items():
    names = collections.Counter(x[0].split("(")[0][:40] for x in v)
    print("stream %s: %4d kernels, from %.3f to %.3f ms, busy %.3f ms | %s" % (k, len(v), (v[0][1] - t0) / 1e6, (max(x[2] for x in v) - t0) / 1e6,
          sum(x[2] - x[1] for x in v) / 1e6, ", ".join("%s x%d" % kv for kv in names.most_common(4))))
if "-v" in sys.argv:  # every kernel of the set-up, in start order
    print("\nstart_ms  dur_us  stream  kernel")
    for r in seg:
        print("%8.3f %7.1f  %6s  %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, r[3] if key else 0, r[0].split("(")[0][:60]))
