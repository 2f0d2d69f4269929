#!/bin/bash
# kernel trace of tools/fine_ab.py at 256^3 (library path) and of the probe (same kernels, synthetic data): durations side by side
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_lib -- python $R/tools/fine_ab.py 256 256 256 > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/prof_ab_lib -name "*.db" | head -n 1) | head -8 | cut -c1-150
find $R/gpurun_out/prof_ab_lib -name "*.db" -delete
export PROBE_ONE_KZ=1 PROBE_FACE_ONLY=1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_probe -- $R/tools/probe/fine_probe 256 256 256 10 43 10 > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/prof_ab_probe -name "*.db" | head -n 1) | head -14 | cut -c1-150
find $R/gpurun_out/prof_ab_probe -name "*.db" -delete
