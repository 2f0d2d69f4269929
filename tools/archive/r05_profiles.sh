#!/bin/bash
# round-5 evidence.  usage: bash tools/r05_profiles.sh [c4] [bench] [cube256] [pmc] [line]   (no argument: everything)
#   line     the DEFAULT bench command's line -> r05_bench_line.json
#   bench    kernel trace of the same command (headline part) -> per-kernel stats, idle gaps, step shares, one CG iteration, set-up streams
#   c4       kernel trace of config 4 (MBB beam, Helmholtz filter) -> per-kernel stats + step shares, and its bench line
#   cube256  kernel trace of the 256^3 fine kernels
#   pmc      HBM traffic at 128^3 / 256^3 (separate --pmc passes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
want() { [ $# -eq 0 ] && return 0; for a in $ARGS; do [ "$a" = "$1" ] && return 0; done; [ -z "$ARGS" ]; }
ARGS="$*"
trace() {  # trace <tag> <bench args...>: kernel trace of a short bench with the micro-measurements shortened
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  ( cd /tmp && TP_BENCH_MEASURE_S=0.02 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 3 --warmup 2 "$@" > $R/gpurun_out/r05_${tag}_prof_line.json 2>/dev/null )
  echo "rocprofv3 $tag rc=$?"
  DB=$(find /tmp/prof_$tag -name "*.db" | head -n 1)
}
if want line; then
  timeout 700 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench rc=$?"
fi
if want bench; then
  trace bench
  if [ -n "$DB" ]; then
    python profiles/summarize_rocpd.py $DB > gpurun_out/r05_bench_kernel_stats.csv
    python tools/gaps.py $DB > gpurun_out/r05_bench_idle_gaps.txt
    python tools/step_shares.py $DB > gpurun_out/r05_bench_step_shares.txt
    python tools/iter_timeline.py $DB > gpurun_out/r05_iteration_timeline.txt
    python tools/setup_trace.py $DB > gpurun_out/r05_setup_streams.txt
  fi
  rm -rf /tmp/prof_bench
  head -n 12 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-130; head -8 gpurun_out/r05_bench_idle_gaps.txt; head -n 24 gpurun_out/r05_bench_step_shares.txt
fi
if want c4; then
  trace c4 --workload c4
  if [ -n "$DB" ]; then
    python profiles/summarize_rocpd.py $DB > gpurun_out/r05_c4_kernel_stats.csv
    python tools/step_shares.py $DB > gpurun_out/r05_c4_step_shares.txt
    python tools/gaps.py $DB > gpurun_out/r05_c4_idle_gaps.txt
  fi
  rm -rf /tmp/prof_c4
  timeout 300 python bench.py --workload c4 --no-cube256 --cpu-budget 60 > gpurun_out/r05_c4_bench_line.json 2>/dev/null; echo "c4 bench rc=$?"
  head -n 30 gpurun_out/r05_c4_step_shares.txt
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c4_bench_line.json"))
print("c4: ms", d["ms_per_step"], "its", d["config"]["cg_its"], "launches", d["config"]["kernel_launches_per_step"], "roofline", {k: (v.get("frac") if isinstance(v, dict) else v) for k, v in d["roofline"].items() if k in ("frac", "pde_filter", "conv_filter", "spmv")})
print("parity", json.dumps(d.get("parity"))[:1500])
PY
fi
if want cube256; then
  rm -rf /tmp/prof_256
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_256 -- python $R/tools/fine_ab.py 256 256 256 20 > /dev/null 2>&1 )
  python profiles/summarize_rocpd.py $(find /tmp/prof_256 -name "*.db" | head -n 1) > gpurun_out/r05_cube256_kernel_stats.csv
  rm -rf /tmp/prof_256
  grep "fine_" gpurun_out/r05_cube256_kernel_stats.csv | cut -c1-130
fi
if want pmc; then
  rm -rf /tmp/pmc_r05
  for n in 128 256; do
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_r05/$n/$c -- python $R/tools/pmc_traffic.py $n $n $n > /dev/null 2>&1 )
    done
    python tools/pmc_extract.py /tmp/pmc_r05/$n $n $n $n > gpurun_out/r05_pmc_traffic_$n.json
  done
  rm -rf /tmp/pmc_r05
  cat gpurun_out/r05_pmc_traffic_128.json gpurun_out/r05_pmc_traffic_256.json | grep -v "^ *\"launches\|calib" | head -60
fi
