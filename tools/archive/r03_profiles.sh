# round-3 evidence: kernel trace + idle gaps + per-kernel shares of ONE design iteration of the DEFAULT bench command,
# kernel trace of the 256^3 fine kernels, PMC HBM traffic at 128^3 / 256^3 (separate --pmc passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r03 gpurun_out/prof_r03_256 gpurun_out/pmc_r03
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03 -- python bench.py --no-cpu-baseline > gpurun_out/r03_bench_prof.json 2>/dev/null
DB=$(find gpurun_out/prof_r03 -name "*.db" | head -n 1)
python profiles/summarize_rocpd.py $DB > gpurun_out/r03_bench_kernel_stats.csv
python tools/gaps.py $DB > gpurun_out/r03_bench_idle_gaps.txt
python tools/step_shares.py $DB > gpurun_out/r03_bench_step_shares.txt
rm -rf gpurun_out/prof_r03
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_256 -- python tools/fine_ab.py 256 256 256 20 > gpurun_out/r03_cube256_fine_ab.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_r03_256 -name "*.db" | head -n 1) > gpurun_out/r03_cube256_kernel_stats.csv
rm -rf gpurun_out/prof_r03_256
for n in 128 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r03/$n/$c -- python tools/pmc_traffic.py $n $n $n > /dev/null 2>&1
  done
  python tools/pmc_extract.py gpurun_out/pmc_r03/$n $n $n $n > gpurun_out/r03_pmc_traffic_$n.json
done
rm -rf gpurun_out/pmc_r03
head -n 12 gpurun_out/r03_bench_kernel_stats.csv | cut -c1-120; grep "fine_" gpurun_out/r03_cube256_kernel_stats.csv | cut -c1-120
cat gpurun_out/r03_bench_idle_gaps.txt; head -n 30 gpurun_out/r03_bench_step_shares.txt
cat gpurun_out/r03_pmc_traffic_128.json gpurun_out/r03_pmc_traffic_256.json
