#!/bin/bash
# GPU-side driver of the fine-operator probe: bit checks on small meshes (odd sizes: every parity case of the 16-byte
# staging windows), then timings at 256^3.  Output -> gpurun_out/fine_probe.log
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
L=../../gpurun_out/fine_probe.log
: > $L
run() { echo "### $@" >> $L; timeout 300 "$@" >> $L 2>&1; echo "### exit $?" >> $L; }
run ./fine_probe 37 29 23 1 5 1
run ./fine_probe 48 24 24 1 7 1
run ./fine_probe 37 29 23 1 5 5
run ./fine_probe 48 24 24 1 7 5
run ./fine_probe 64 64 64 1 16 5
run ./fine_probe 256 256 256 10 16 2
run ./fine_probe 256 256 256 10 16 6
run ./fine_probe_skel 256 256 256 10 16 6
tail -n 150 $L
