# ksp_mode 1 (the reference's hard-coded FGMRES / GMRES / SOR configuration) on the GPU box: parity tests + timing at 64^3
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_refksp.py -q -m gpu --tb=short > gpurun_out/refksp_tests.log 2>&1
echo "rc $?" >> gpurun_out/refksp_tests.log
timeout 500 python -m pytest tests/test_reference_on_shim.py -q -m gpu --tb=short -k "hard_coded or unsupported" >> gpurun_out/refksp_tests.log 2>&1
echo "rc $?" >> gpurun_out/refksp_tests.log
tail -n 70 gpurun_out/refksp_tests.log
