#!/bin/bash
# kernel trace of the default bench (headline part only) -> kernel-by-kernel timeline of one CG iteration
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/it && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/it -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cube256 --no-stated-cycle --steps 3 --warmup 2 > /tmp/it.log 2>&1
db=$(find /tmp/it -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/iter_timeline.py $db
