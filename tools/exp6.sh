export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -n 8
echo "ablation 9"; TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl9.so timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
TP_LIB=$PWD/topopt_in_petsc_amd/libtopopt_abl9.so TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
echo base
timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
TP_TILE_KZ=16 timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1
TP_TILE_KZ=32 timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1
