export TMPDIR=/tmp TP_BENCH_TRACE=1 TP_BENCH_FAULT=40
mkdir -p gpurun_out/hang
for i in 1 2 3; do
  timeout -k 5 420 python -m pytest tests/test_bench_line.py -m gpu -x -q > gpurun_out/hang/pytest$i.log 2>&1; echo "pytest $i rc=$?"; tail -n 3 gpurun_out/hang/pytest$i.log
done
