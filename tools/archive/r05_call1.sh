#!/bin/bash
# round 5, GPU call 1: the new parity object (arbiter + converged step) at the metric mesh, the same with the Chebyshev
# coarse solve (is the exact coarse solve what separates GPU and oracle?), TP_CG_NT A/B, the new tests, a trace of config 4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r05_line_a.json 2> gpurun_out/r05_line_a.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_line_a.json"))
print("ms", d["ms_per_step"], "its", d["config"]["cg_its"], "frac", d["roofline"]["frac"], "launches", d["config"]["kernel_launches_per_step"])
print(json.dumps(d.get("parity"), indent=1))
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["seconds"], (cb.get("extras") or {}).get("seconds"))
PY
TP_CPU_THREADS=64 OMP_PROC_BIND=spread OMP_PLACES=cores timeout 400 python bench.py --coarse cheb --no-cube256 --no-stated-cycle --steps 3 > gpurun_out/r05_line_cheb.json 2> gpurun_out/r05_line_cheb.err; echo "bench cheb rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_line_cheb.json"))
print("CHEB COARSE: ms", d["ms_per_step"], "its", d["config"]["cg_its"])
print(json.dumps(d.get("parity"), indent=1))
PY
for nt in 0 1 0 1; do
  TP_CG_NT=$nt timeout 200 python bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('NT=$nt ms %.3f frac %.4f avg_launch_ms %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))"
done
timeout 900 python -m pytest tests/test_bench_line.py tests/test_multirank.py::test_give_up_on_one_rank_is_handled_by_all tests/test_gpu_parity.py::test_one_xcd_kernels_give_up_path tests/test_multirank.py::test_two_ranks_one_gpu -x -q -m gpu 2>&1 | tail -15
bash tools/prof_workload.sh c4 2>&1 | tail -20
cp gpurun_out/k_c4.csv gpurun_out/r05_c4_kernel_stats.csv
