# usage: bash tools/prof_bench.sh <tag> [bench args]  -> gpurun_out/k_<tag>.csv (per-kernel totals), gpurun_out/b_<tag>.json
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
python tools/gaps.py $(find gpurun_out/prof_$tag -name "*.db" | head -n 1) > gpurun_out/gaps_$tag.txt; rm -rf gpurun_out/prof_$tag
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/b_$tag.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_$tag -name "*.db" | head -n 1) > gpurun_out/k_$tag.csv
python tools/gaps.py $(find gpurun_out/prof_$tag -name "*.db" | head -n 1) > gpurun_out/gaps_$tag.txt; rm -rf gpurun_out/prof_$tag
head -n 32 gpurun_out/k_$tag.csv | cut -c1-140
