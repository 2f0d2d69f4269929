#!/bin/bash
# round-4 evidence: the DEFAULT bench command's line; kernel trace of the same command (headline part) reduced to per-kernel
# stats, idle gaps, per-kernel shares of one design iteration, one CG iteration kernel by kernel, the set-up per stream;
# kernel trace of the 256^3 fine kernels; PMC HBM traffic at 128^3 / 256^3 (separate --pmc passes).  Every step bounded.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 500 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; echo "bench rc=$?"
rm -rf /tmp/prof_r04
( cd /tmp && TP_BENCH_MEASURE_S=0.02 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04 -- python $R/bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 3 --warmup 2 > $R/gpurun_out/r04_bench_prof.json 2>/dev/null )
echo "rocprofv3 rc=$?"
DB=$(find /tmp/prof_r04 -name "*.db" | head -n 1)
if [ -n "$DB" ]; then
  python profiles/summarize_rocpd.py $DB > gpurun_out/r04_bench_kernel_stats.csv
  python tools/gaps.py $DB > gpurun_out/r04_bench_idle_gaps.txt
  python tools/step_shares.py $DB > gpurun_out/r04_bench_step_shares.txt
  python tools/iter_timeline.py $DB > gpurun_out/r04_iteration_timeline.txt
  python tools/setup_trace.py $DB > gpurun_out/r04_setup_streams.txt
fi
rm -rf /tmp/prof_r04 /tmp/st
( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d /tmp/st -- python $R/tools/r04_setup.py 4 > /dev/null 2>&1 )
python tools/setup_trace3.py $(find /tmp/st -name "*.db" | head -1) > gpurun_out/r04_setup_timeline.txt
rm -rf /tmp/st /tmp/prof_r04_256
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04_256 -- python $R/tools/fine_ab.py 256 256 256 20 > /dev/null 2>&1 )
python profiles/summarize_rocpd.py $(find /tmp/prof_r04_256 -name "*.db" | head -n 1) > gpurun_out/r04_cube256_kernel_stats.csv
rm -rf /tmp/prof_r04_256 /tmp/pmc_r04
for n in 128 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_r04/$n/$c -- python $R/tools/pmc_traffic.py $n $n $n > /dev/null 2>&1 )
  done
  python tools/pmc_extract.py /tmp/pmc_r04/$n $n $n $n > gpurun_out/r04_pmc_traffic_$n.json
done
rm -rf /tmp/pmc_r04
head -n 14 gpurun_out/r04_bench_kernel_stats.csv | cut -c1-130
head -8 gpurun_out/r04_bench_idle_gaps.txt; head -n 26 gpurun_out/r04_bench_step_shares.txt; cat gpurun_out/r04_setup_streams.txt
grep "fine_" gpurun_out/r04_cube256_kernel_stats.csv | cut -c1-130
cat gpurun_out/r04_pmc_traffic_128.json gpurun_out/r04_pmc_traffic_256.json | grep -v "^ *\"launches\|calib" | head -60
