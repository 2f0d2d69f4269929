#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe2.log
: > $L
run() { echo "### $@" >> $L; timeout 300 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
for p in pol1 pol2; do
run ./fine_probe_$p 37 29 23 1 5 1
run ./fine_probe_$p 48 24 24 1 7 1
run ./fine_probe_$p 64 64 64 1 16 5
done
run ./fine_probe_pol2 256 256 256 10 16 6
run ./fine_probe_abl8 256 256 256 10 16 6
run ./fine_probe_abl7 256 256 256 10 16 2
grep -v "bit-identical" $L | tail -n 200
