#!/bin/bash
# strong scaling of the metric mesh on 2, 4, 8 slabs of one GPU with the coarse levels replicated from level 2 / automatically / coarsest only
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 2 8; do
  for rep in 0 "" 2; do
    TP_REPLICATE_FROM=$rep timeout 600 python bench.py --gpus $n --same-device --backend gloo --scaling strong --steps 1 --warmup 1 --budget-s 500 --no-other-scaling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('N=$n strong, TP_REPLICATE_FROM=\"$rep\": its %s coarse %s launches %s overlapped halos %s' % (c.get('cg_its'), c.get('coarse_solve'), c.get('kernel_launches_per_step'), c.get('halo_overlap')))" 
  done
done
