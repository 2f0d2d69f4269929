#!/bin/bash
# round-3 evidence after the one-XCD runs and the exact coarse solve: kernel trace of the DEFAULT bench command (headline
# part), per-kernel stats, idle gaps, per-kernel shares of one design iteration, one CG iteration kernel by kernel, and the
# per-stream timeline of the set-up phase.  Bounded: a trace that does not finish in 200 s is abandoned.
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_r03b
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r03b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03b_bench_prof.json 2>/dev/null )
echo "rocprofv3 rc=$?"
DB=$(find /tmp/prof_r03b -name "*.db" | head -n 1)
[ -z "$DB" ] && { echo "no trace"; exit 1; }
python profiles/summarize_rocpd.py $DB > gpurun_out/r03b_bench_kernel_stats.csv
python tools/gaps.py $DB > gpurun_out/r03b_bench_idle_gaps.txt
python tools/step_shares.py $DB > gpurun_out/r03b_bench_step_shares.txt
python tools/iter_timeline.py $DB > gpurun_out/r03b_iteration_timeline.txt
python tools/setup_trace.py $DB > gpurun_out/r03b_setup_streams.txt
head -n 14 gpurun_out/r03b_bench_kernel_stats.csv | cut -c1-130
cat gpurun_out/r03b_bench_idle_gaps.txt | head -8; head -n 24 gpurun_out/r03b_bench_step_shares.txt; cat gpurun_out/r03b_setup_streams.txt
