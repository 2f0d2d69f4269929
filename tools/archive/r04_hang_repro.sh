# reproduce the round-3 hang of the 2-rank same-device bench at exit; every run under its own timeout
export TMPDIR=/tmp TP_BENCH_TRACE=1 TP_BENCH_FAULT=40
mkdir -p gpurun_out/hang
B="python bench.py --gpus 2 --same-device --backend gloo --workload tiny --steps 2 --warmup 1"
for i in $(seq 1 24); do
  s=weak; [ $((i % 2)) = 0 ] && s=strong
  t0=$(date +%s.%N)
  if [ $((i % 3)) = 0 ]; then
    timeout -k 5 90 $B --scaling $s 2>&1 | cat > gpurun_out/hang/run$i.log; rc=${PIPESTATUS[0]}
  else
    timeout -k 5 90 $B --scaling $s > gpurun_out/hang/run$i.log 2> gpurun_out/hang/run$i.err; rc=$?
  fi
  t1=$(date +%s.%N)
  echo "run $i $s rc=$rc $(echo "$t1 - $t0" | bc) s"
  if [ $rc != 0 ]; then echo "HANG/FAIL at $i"; tail -n 60 gpurun_out/hang/run$i.* | cut -c1-400; break; fi
  rm -f gpurun_out/hang/run$i.*
done
