#!/bin/bash
# round-6 evidence.  usage: bash tools/r06_profiles.sh [bench] [cube256] [pmc] [pmc2] [c4]   (no argument: everything)
#   bench    kernel trace of the DEFAULT bench command (headline part) -> per-kernel stats, idle gaps, step shares, one CG iteration, set-up streams
#   cube256  kernel trace of the 256^3 fine kernels
#   pmc      HBM traffic of the fine kernels at 128^3 / 256^3 (separate --pmc passes; refreshes profiles/spmv_traffic.json's source)
#   pmc2     counters of the kernels nobody had looked at: cone filter, 0 <-> 1 transfers, level-2 stencil, Krylov product
#   c4       config 4 lines: as configured, and with the Helmholtz filter solved to rtol 1e-13 on both sides
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
ARGS="$*"
want() { [ -z "$ARGS" ] && return 0; for a in $ARGS; do [ "$a" = "$1" ] && return 0; done; return 1; }
trace() {  # trace <tag> <bench args...>
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  ( cd /tmp && TP_BENCH_MEASURE_S=0.02 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --design-loop 0 --steps 3 --warmup 2 "$@" > $R/gpurun_out/r06_${tag}_prof_line.json 2>/dev/null )
  echo "rocprofv3 $tag rc=$?"
  DB=$(find /tmp/prof_$tag -name "*.db" | head -n 1)
}
if want bench; then
  trace bench
  if [ -n "$DB" ]; then
    python profiles/summarize_rocpd.py $DB > gpurun_out/r06_bench_kernel_stats.csv
    python tools/gaps.py $DB > gpurun_out/r06_bench_idle_gaps.txt
    python tools/step_shares.py $DB > gpurun_out/r06_bench_step_shares.txt
    python tools/iter_timeline.py $DB > gpurun_out/r06_iteration_timeline.txt
    python tools/setup_trace.py $DB > gpurun_out/r06_setup_streams.txt
    python tools/setup_trace.py $DB -v > gpurun_out/r06_setup_timeline.txt
  fi
  rm -rf /tmp/prof_bench
  head -n 14 gpurun_out/r06_bench_kernel_stats.csv | cut -c1-130; head -8 gpurun_out/r06_bench_idle_gaps.txt; head -n 26 gpurun_out/r06_bench_step_shares.txt
fi
if want cube256; then
  rm -rf /tmp/prof_256
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_256 -- python $R/tools/fine_ab.py 256 256 256 20 > /dev/null 2>&1 )
  python profiles/summarize_rocpd.py $(find /tmp/prof_256 -name "*.db" | head -n 1) > gpurun_out/r06_cube256_kernel_stats.csv
  rm -rf /tmp/prof_256
  grep "fine_" gpurun_out/r06_cube256_kernel_stats.csv | cut -c1-130
fi
if want pmc; then
  rm -rf /tmp/pmc_r06
  for n in 128 256; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_r06/$n/$c -- python $R/tools/pmc_traffic.py $n $n $n > /dev/null 2>&1 )
    done
    python tools/pmc_extract.py /tmp/pmc_r06/$n $n $n $n > gpurun_out/r06_pmc_traffic_$n.json
  done
  rm -rf /tmp/pmc_r06
  cat gpurun_out/r06_pmc_traffic_128.json gpurun_out/r06_pmc_traffic_256.json | grep -v "^ *\"launches\|calib" | head -60
fi
if want pmc2; then
  rm -rf /tmp/pmc2_r06
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc2_r06/s$i -- python $R/tools/r06_pmc_kernels.py 128 > /dev/null 2>&1 ); echo "pmc2 pass $i rc=$?"
  done
  python tools/r06_pmc_reduce.py /tmp/pmc2_r06 128 > gpurun_out/r06_pmc_counters_other_kernels.json
  rm -rf /tmp/pmc2_r06
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_pmc_counters_other_kernels.json"))
for k, e in d["kernels"].items():
    print(k, "dur %.1f us" % e["dur_us_profiled"], "traffic/alg %s" % e.get("traffic_over_algorithmic"), "shares", {a: round(b, 3) for a, b in (e.get("wave_time_shares") or {}).items()})
    print("    instr/wave", {a.replace("SQ_INSTS_", ""): round(b, 1) for a, b in (e.get("instructions_per_wave") or {}).items()})
PY
fi
if want c4; then
  timeout 600 python bench.py --workload c4 --cpu-budget 1200 --no-cube256 > gpurun_out/r06_c4_line.json 2> gpurun_out/r06_c4_line.err; echo "c4 rc=$?"
  timeout 600 python bench.py --workload c4 --cpu-budget 1200 --no-cube256 --pde-rtol 1e-13 > gpurun_out/r06_c4_tight_line.json 2> gpurun_out/r06_c4_tight_line.err; echo "c4 tight rc=$?"
  python - <<'PY'
import json
for f in ("r06_c4_line", "r06_c4_tight_line"):
    d = json.load(open("gpurun_out/%s.json" % f)); p = d["parity"]
    print(f, "ms", d["ms_per_step"], "its", d["config"]["cg_its"], "ok", p["ok"], p["breaches"], "fx", p["fx_rel_err"], "hist10", p["hist_max_rel_err_first10"], "all", p["hist_max_rel_err_all"])
    print("   ", json.dumps(d["roofline"]["pde_filter"].get("solver_comparison")))
PY
fi
