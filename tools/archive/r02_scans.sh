# round-2 one-off scans, kept in one file for the record (each block was one gpurun call; tools/README.md)

# ---- was tools/exp1.sh
if [ "$1" = "exp1" ]; then
  export TMPDIR=/tmp
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -n 5
  for v in 1 2; do for n in 128 256; do
  TP_FINE_V=$v timeout 120 python tools/fine_ab.py $n $n $n 2>&1 | tail -n 1
  done; done
  for kz in 6 12 16; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1; done
  for kz in 16 32; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1; done
  for v in 1 2; do TP_FINE_V=$v timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v$v ms_per_step', d['ms_per_step'], 'its', d['config']['cg_its'], 'cheb_ms', d['roofline']['avg_launch_ms'], 'spmv_ms', d['roofline']['spmv']['avg_launch_ms'])"; done
fi

# ---- was tools/exp3.sh
if [ "$1" = "exp3" ]; then
  export TMPDIR=/tmp
  for a in 1 2 3 4 5 6; do
  echo "ablation $a"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  done
  echo baseline; timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  bash tools/pmc_sq.sh
fi

# ---- was tools/exp4.sh
if [ "$1" = "exp4" ]; then
  export TMPDIR=/tmp
  for a in 7 8; do
  echo "ablation $a"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl$a.so TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
  done
  echo baseline; TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
  rm -rf gpurun_out/pmc_clk
  timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python tools/pmc_traffic.py 128 128 128 > /dev/null 2>&1
  python - <<'PY'
  import csv, glob, collections
  cnt = collections.defaultdict(list); dur = collections.defaultdict(list)
  for fn in glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True):
      for r in csv.DictReader(open(fn)):
          cnt[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
  for fn in glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv", recursive=True):
      for r in csv.DictReader(open(fn)):
          dur[r["Kernel_Name"][:40]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
  for k in cnt:
      if "tile" in k or "k_scale" in k:
          c = sum(cnt[k]) / len(cnt[k]); d = sum(dur[k]) / len(dur[k])
          print(k, "GUI_ACTIVE", round(c), "dur_ns", round(d), "GHz", round(c / d, 3))
  PY
fi

# ---- was tools/exp5.sh
if [ "$1" = "exp5" ]; then
  export TMPDIR=/tmp
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -n 8
  for v in 1 2; do 
  TP_FINE_V=$v timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  TP_FINE_V=$v timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
  done
  for kz in 16; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1; done
  for kz in 32; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1; done
  for v in 1 2; do TP_FINE_V=$v timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v$v ms_per_step', d['ms_per_step'], 'its', d['config']['cg_its'], 'cheb_ms', d['roofline']['avg_launch_ms'], 'spmv_ms', d['roofline']['spmv']['avg_launch_ms'])"; done
fi

# ---- was tools/exp6.sh
if [ "$1" = "exp6" ]; then
  export TMPDIR=/tmp
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -n 8
  echo "ablation 9"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl9.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl9.so TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
  echo base
  timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  TP_TILE_KZ=16 timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
  TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
fi

# ---- was tools/exp7.sh
if [ "$1" = "exp7" ]; then
  export TMPDIR=/tmp
  P="FETCH_SIZE WRITE_SIZE TCC_HIT_sum+TCC_MISS_sum+TCC_REQ_sum+TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum+TCP_TCC_WRITE_REQ_sum+TCP_TOTAL_CACHE_ACCESSES_sum+TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum+TCP_UTCL1_TRANSLATION_HIT_sum+TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum+TCP_TCP_TA_DATA_STALL_CYCLES_sum+TA_ADDR_STALLED_BY_TC_CYCLES_sum+TCP_TCR_TCP_STALL_CYCLES_sum SQ_WAVE_CYCLES+SQ_WAIT_ANY+SQ_WAIT_INST_ANY+SQ_ACTIVE_INST_ANY+SQ_ACTIVE_INST_VALU+SQ_ACTIVE_INST_VMEM+SQ_ACTIVE_INST_LDS+SQ_BUSY_CYCLES"
  TP_TILE_KZ=32 python tools/pmc_multi.py gpurun_out/pmc256 256 $P > gpurun_out/pmc256_v2.json
  TP_TILE_KZ=32 TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl9.so python tools/pmc_multi.py gpurun_out/pmc256s 256 $P > gpurun_out/pmc256_skel.json
  rm -rf gpurun_out/pmc256 gpurun_out/pmc256s
  cat gpurun_out/pmc256_v2.json gpurun_out/pmc256_skel.json
fi

# ---- was tools/exp8.sh
if [ "$1" = "exp8" ]; then
  export TMPDIR=/tmp
  for kz in 8 15 22 33; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1; done
  for kz in 16 32 65 129; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1; done
  for kz in 4 5 6 8 11 13; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 64 64 2>&1 | tail -n 1; done
  for kz in 4 6 8 11 16; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 192 64 64 2>&1 | tail -n 1; done
  for kz in 8 16 26 43 65; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 128 128 2>&1 | tail -n 1; done
fi

# ---- was tools/sweep2.sh
if [ "$1" = "sweep2" ]; then
  export TMPDIR=/tmp
  mkdir -p gpurun_out
  bash tools/sweep_solver_params.sh "c1 c3 c4" "2 4" "30 60" > /dev/null
  cp gpurun_out/sweep_params.txt gpurun_out/sweep_params_b.txt
  for nl in 5; do for nc in 30 60; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse $nc --nsmooth 2 --nlvls $nl 2>/dev/null | python -c "
  import json,sys
  d=json.loads(sys.stdin.readline()); c=d['config']
  print('cantilever128 nlvls $nl nsmooth 2 ncoarse $nc : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
  done; done
  timeout 300 python bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse 45 --nsmooth 2 --nlvls 4 2>/dev/null | python -c "
  import json,sys
  d=json.loads(sys.stdin.readline()); c=d['config']
  print('c2 nlvls 4 nsmooth 2 ncoarse 45 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
  timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse 60 --nsmooth 2 2>/dev/null | python -c "
  import json,sys
  d=json.loads(sys.stdin.readline()); c=d['config']
  print('c5 nsmooth 2 ncoarse 60 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> gpurun_out/sweep_params_b.txt
  cat gpurun_out/sweep_params_b.txt
fi

# ---- was tools/sweep3.sh
if [ "$1" = "sweep3" ]; then
  export TMPDIR=/tmp
  mkdir -p gpurun_out
  out=gpurun_out/sweep_params_c.txt
  : > $out
  run() {  # workload nlvls nsmooth ncoarse steps
  timeout 300 python bench.py --workload $1 --steps $5 --warmup 1 --no-cpu-baseline --no-cube256 --ncoarse $4 --nsmooth $3 --nlvls $2 2>/dev/null | python -c "
  import json,sys
  d=json.loads(sys.stdin.readline()); c=d['config']
  print('$1 nlvls $2 nsmooth $3 ncoarse $4 : %.2f ms/step, CG its %s' % (d['ms_per_step'], c.get('cg_its')))" >> $out
  }
  for nl in 5 6; do for nc in 16 24 30 45; do run cantilever128 $nl 2 $nc 5; done; done
  run cantilever128 5 1 30 5
  run cantilever128 5 3 30 5
  run c1 4 2 16 5
  run c1 4 2 22 5
  run c3 5 2 30 5
  run c3 5 2 45 5
  run c4 4 2 30 5
  run c4 4 2 45 5
  run c5 5 2 45 2
  run c5 6 2 30 2
  run c2 3 2 45 5
  cat $out
fi
