#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(time timeout 1200 python -m pytest tests/test_multirank.py tests/test_bench_line.py -x -q -m gpu 2>&1 | tail -15) 2>&1 | tail -20
# how many halos travel on the second stream now (2 slabs, bench cycle)
timeout 300 python bench.py --gpus 2 --same-device --backend gloo --steps 1 --warmup 1 --no-other-scaling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('overlapped halos', c['halo_overlap'], 'its', c['cg_its'], 'launches', c['kernel_launches_per_step'])"
TP_STENCIL_OVERLAP=0 timeout 300 python bench.py --gpus 2 --same-device --backend gloo --steps 1 --warmup 1 --no-other-scaling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('TP_STENCIL_OVERLAP=0: overlapped halos', c['halo_overlap'], 'its', c['cg_its'], 'launches', c['kernel_launches_per_step'])"
