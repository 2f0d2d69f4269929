#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe9.log
: > $L
run() { echo "### $@" >> $L; timeout 120 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
export PROBE_ONE_KZ=1
run ./fine_probe 37 29 23 1 5 13
run ./fine_probe 129 65 33 1 8 13
for kz in 8 16 33; do run ./fine_probe 256 128 128 20 $kz 10; done
for kz in 4 8; do run ./fine_probe 192 64 64 20 $kz 10; done
for kz in 2 4; do run ./fine_probe 48 24 24 50 $kz 10; done
for kz in 16 29 43; do run ./fine_probe 512 256 256 5 $kz 10; done
grep -v "bit-identical\|stream" $L | head -n 150
