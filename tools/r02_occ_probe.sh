#!/bin/bash
# occupancy experiment on the BVH path: builds of the BVH kernels for 4, 5 and 6 blocks per CU (build_ab/, -DPT_BVH_WAVES=n)
OUT=gpurun_out/occ_probe.txt; : > $OUT
for pipe in mega wavefront; do
for w in 4 5 6; do
  lib=$PWD/build_ab/libptcore_w$w.so; [ $w = 4 ] && lib=$PWD/path_trace_golang_amd/libptcore.so
  echo "== PTCORE_PIPELINE=$pipe waves $w" >> $OUT
  PTCORE_VERBOSE=1 PTCORE_PIPELINE=$pipe PTCORE_LIB=$lib timeout -k 10 120 python tools/probe_synth.py ${1:-100000} >> $OUT 2>&1 || exit 1
done
done
