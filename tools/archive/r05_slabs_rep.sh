#!/bin/bash
# strong scaling of the metric mesh on 2, 4, 8 slabs of one GPU: coarse levels replicated from level 2 (TP_REPLICATE_FROM=2) / by the
# automatic rule (unset) / coarsest level only (0) -- iteration counts, launches, halos that travelled behind computation
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --gpus $1 --same-device --backend gloo --scaling strong --steps 1 --warmup 1 --budget-s 500 --no-other-scaling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('N=$1 strong, $2: its %s coarse %s launches %s overlapped halos %s' % (c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step'), c.get('halo_overlap')))"; }
for n in 2 4 8; do
  unset TP_REPLICATE_FROM; run $n "automatic"
  TP_REPLICATE_FROM=0 run $n "coarsest only"
  TP_REPLICATE_FROM=2 run $n "from level 2"
done
