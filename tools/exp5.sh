export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -n 8
for v in 1 2; do 
TP_FINE_V=$v timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
TP_FINE_V=$v timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
done
for kz in 16; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1; done
for kz in 32; do TP_FINE_V=2 TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1; done
for v in 1 2; do TP_FINE_V=$v timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('v$v ms_per_step', d['ms_per_step'], 'its', d['config']['cg_its'], 'cheb_ms', d['roofline']['avg_launch_ms'], 'spmv_ms', d['roofline']['spmv']['avg_launch_ms'])"; done
