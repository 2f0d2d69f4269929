# the round's GPU acceptance: full -m gpu suite (no -x: one failure must not hide the rest), smoke, default bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -q -m gpu --durations=25 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -n 45 gpurun_out/pytest.log; tail -n 2 gpurun_out/smoke.log; cat gpurun_out/bench_default.json; tail -n 5 gpurun_out/bench_default.err
