#!/bin/bash
# builds tools/probe/fine_probe (all variants) and fine_probe_skel (-DFD_ABL=9: data movement only); gfx950, no GPU needed
cd "$(dirname "$0")"
F="-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $F -o fine_probe fine_probe.hip 2>&1 | grep -v "inline asm clobber\|^note\|warnings\? generated" &
/opt/rocm/bin/hipcc $F -DFD_ABL=9 -o fine_probe_skel fine_probe.hip 2>&1 | grep -v "inline asm clobber\|^note\|warnings\? generated" &
wait
ls -la fine_probe fine_probe_skel
