#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe6.log
: > $L
run() { echo "### $@" >> $L; timeout 120 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
export PROBE_ONE_KZ=1
for kz in 8 11 15 22; do run ./fine_probe 128 128 128 20 $kz 6; done
for kz in 4 8; do run ./fine_probe 128 64 64 20 $kz 6; done
for kz in 8 16; do run ./fine_probe 256 128 128 20 $kz 6; done
grep -v "bit-identical\|stream" $L | head -n 190
