# round-2 evidence: kernel-trace summaries of the bench and of the 256^3 fine kernels, PMC HBM traffic at 128^3 / 256^3
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r02 gpurun_out/prof_r02_256 gpurun_out/pmc_r02
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cube256 > gpurun_out/r02_bench_prof.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_r02 -name "*.db" | head -n 1) > gpurun_out/r02_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r02
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02_256 -- python tools/fine_ab.py 256 256 256 20 > gpurun_out/r02_cube256_fine_ab.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_r02_256 -name "*.db" | head -n 1) > gpurun_out/r02_cube256_kernel_stats.csv
rm -rf gpurun_out/prof_r02_256
for n in 128 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r02/$n/$c -- python tools/pmc_traffic.py $n $n $n > /dev/null 2>&1
  done
  python tools/pmc_extract.py gpurun_out/pmc_r02/$n $n $n $n > gpurun_out/r02_pmc_traffic_$n.json
done
rm -rf gpurun_out/pmc_r02
head -n 14 gpurun_out/r02_bench_kernel_stats.csv | cut -c1-120; grep fine_tile gpurun_out/r02_cube256_kernel_stats.csv | cut -c1-120
cat gpurun_out/r02_pmc_traffic_128.json gpurun_out/r02_pmc_traffic_256.json
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; cat gpurun_out/r02_bench_default.json
