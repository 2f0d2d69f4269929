#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe10.log
: > $L
run() { echo "### $@" >> $L; timeout 120 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
export PROBE_ONE_KZ=1
run ./fine_probe 256 256 256 10 43 10
run ./fine_probe 128 128 128 20 15 10
export PROBE_FACE_ONLY=1
echo "### face mask" >> $L
run ./fine_probe 256 256 256 10 43 10
run ./fine_probe 128 128 128 20 15 10
grep -v "bit-identical\|stream" $L | head -n 150
