#!/bin/bash
# same-box A/B of two builds of libptcore.so on the BVH probe: tools/ab_probe.sh <libA> <libB> [objects ...]
A=$1; B=$2; shift 2
OUT=gpurun_out/ab_probe.txt; : > $OUT
for rep in 1 2; do
for lib in $A $B; do
  echo "== $lib (run $rep)" >> $OUT
  PTCORE_LIB=$PWD/$lib timeout -k 10 200 python tools/probe_synth.py ${@:-100000} 2>&1 | grep -v amdgpu.ids >> $OUT || exit 1
done
done
