for sh in 2 3; do echo "TP_FINE_SHAPE=$sh"; TP_FINE_SHAPE=$sh python tools/cheb_variants.py 256 2>&1 | grep copy-only; done
for kz in 22 33 65; do echo "TP_FINE_SHAPE=3 TP_TILE_KZ=$kz"; TP_FINE_SHAPE=3 TP_TILE_KZ=$kz python tools/cheb_variants.py 256 2>&1 | grep copy-only; done
TP_FINE_SHAPE=3 TP_FINE_V=3 python -m pytest tests/test_gpu_fine_generations.py -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed" gpurun_out/t.log | tail -1
