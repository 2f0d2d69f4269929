export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_asm -- python $R/tools/asm_only.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob, collections
db = glob.glob("$R/gpurun_out/prof_asm/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# last assemble = from the last k_simp-like first kernel: split by the first kernel name of an assemble
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_galerkin_l2_fast") or r[0].startswith("k_galerkin_fine")]
# one assemble = from one first-Galerkin kernel to the next
firsts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 50]
seg = rows[firsts[-2]:firsts[-1]]
t0, t1 = seg[0][1], seg[-1][2]
agg = collections.OrderedDict()
for n, s, e in seg:
    a = agg.setdefault(n.split("(")[0][:70], [0, 0]); a[0] += 1; a[1] += e - s
busy = sum(a[1] for a in agg.values())
print("one assemble: span %.3f ms, %d kernels, sum of kernel times %.3f ms" % ((t1 - t0) / 1e6, len(seg), busy / 1e6))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%9.1f us  x%-4d %s" % (a[1] / 1e3, a[0], n))
PY
rm -rf $R/gpurun_out/prof_asm
