export TMPDIR=/tmp
for z in 0 2 4; do echo "TP_FILTER_ZMULTI=$z"; TP_FILTER_ZMULTI=$z timeout 300 python tools/r06_filter_ab.py 2>&1 | grep ElemConn; done
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob("/tmp/filt_*_0.npy")):
    a = np.load(f)
    for z in ("2", "4"):
        b = np.load(f.replace("_0.npy", "_%s.npy" % z))
        print(f, z, "bitwise equal" if np.array_equal(a, b) else "DIFFERENT max %.3e" % np.abs(a - b).max())
PY
