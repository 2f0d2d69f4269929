set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
rm -rf gpurun_out/prof_final gpurun_out/pmc_final
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
timeout 120 python profiles/summarize_rocpd.py $(find gpurun_out/prof_final -name '*.db' | head -n 1) > gpurun_out/final_kernel_stats.csv 2> gpurun_out/summarize.err
find gpurun_out/prof_final -name "*.db" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_final/$c -- python tools/pmc_traffic.py 128 128 128 > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_extract.py gpurun_out/pmc_final 128 128 128 > gpurun_out/traffic_128.json 2> gpurun_out/pmc_extract.err
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -n 3 gpurun_out/pytest.log; cat gpurun_out/smoke.log | tail -n 2; cat gpurun_out/bench_prof.json; head -n 12 gpurun_out/final_kernel_stats.csv; cat gpurun_out/traffic_128.json; cat gpurun_out/bench_default.json
