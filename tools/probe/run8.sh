#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe8.log
: > $L
run() { echo "### $@" >> $L; timeout 120 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
export PROBE_ONE_KZ=1
run ./fine_probe 37 29 23 1 5 9
run ./fine_probe 129 65 33 1 8 9
for kz in 29 43 65; do run ./fine_probe 256 256 256 10 $kz 10; done
run ./fine_probe 128 128 128 20 15 10
run ./fine_probe 128 128 128 20 22 10
grep -v "bit-identical\|stream" $L | head -n 150
