cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/r04_setup.py 2>&1 | grep "set-up"
TP_CD_INVERT_COLUMNS=1 python $R/tools/r04_setup.py 2>&1 | grep "set-up"
cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "coarsest_level or solve_residual or give_up or bench_cycle" > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t.log | tail -3
