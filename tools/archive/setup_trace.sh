#!/bin/bash
# kernel trace of a few set-up phases (tools/setup_time.py) -> per-stream timeline of the last one
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st -- python $GRAFT_REPO_ROOT/tools/phases.py > /tmp/st.log 2>&1
db=$(find /tmp/st -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/setup_trace.py $db
