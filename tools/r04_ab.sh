run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ms %.3f its %d rel %.6e fx %.12e launches %d' % (d['ms_per_step'], d['config']['cg_its'], d['config']['rel_residual'], d['config']['fx'], d['config']['kernel_launches_per_step']))"; }
run TP_NO_CG_FUSE=1
run A=1
run TP_NO_CG_FUSE=1
run A=1
