#!/bin/bash
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe5.log
: > $L
run() { echo "### $@" >> $L; timeout 120 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
run ./fine_probe 37 29 23 1 5 5
run ./fine_probe 48 24 24 1 7 5
run ./fine_probe 129 65 33 1 8 5
run ./fine_probe 256 256 256 10 16 6
grep -v "bit-identical" $L | head -n 150
