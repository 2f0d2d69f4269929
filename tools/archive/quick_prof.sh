# usage: bash tools/quick_prof.sh <tag> [pattern]   -- parity tests + kernel-trace of the bench, prints the main kernels
export TMPDIR=/tmp
tag=$1; pat=${2:-tile}
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_$tag.json 2>/dev/null
python profiles/summarize_rocpd.py $(find gpurun_out/prof_$tag -name "*.db" | head -n 1) > gpurun_out/k_$tag.csv
find gpurun_out/prof_$tag -name "*.db" -delete
grep "$pat" gpurun_out/k_$tag.csv | cut -c1-120
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'its', d['config']['cg_its'], 'cheb_ms', d['roofline']['avg_launch_ms'])"
