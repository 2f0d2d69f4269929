#!/bin/bash
# A/B of the fine-operator generations through the library (tools/fine_ab.py): TP_FINE_V=2 (k_fine_tile) vs 3 (fine_u4.h)
export TMPDIR=/tmp
MESHES=${MESHES:-"256,256,256 128,128,128 256,128,128 128,64,64"}
for mesh in $MESHES; do
for v in "TP_FINE_V=2" "TP_FINE_V=3" "TP_FINE_V=3 TP_FINE_SHAPE=1" "TP_FINE_V=3 TP_FINE_SHAPE=2"; do
env $v timeout 300 python tools/fine_ab.py ${mesh//,/ } 2>&1 | tail -n 1
done; done
