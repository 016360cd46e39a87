#!/bin/bash
# occupancy experiment on the BVH path: builds of the BVH kernels for 4, 5 and 6 blocks per CU.  Build the variants first (in the
# container, they travel with the snapshot):
#   mkdir -p build_ab && for w in 5 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC \
#     -DPT_BVH_WAVES=$w -Iinclude -shared path_trace_golang_amd/csrc/ptcore.hip -o build_ab/libptcore_w$w.so; done
OUT=gpurun_out/occ_probe.txt; : > $OUT
for pipe in mega wavefront; do
for w in 4 5 6; do
  lib=$PWD/build_ab/libptcore_w$w.so; [ $w = 4 ] && lib=$PWD/path_trace_golang_amd/libptcore.so
  echo "== PTCORE_PIPELINE=$pipe waves $w" >> $OUT
  PTCORE_VERBOSE=1 PTCORE_PIPELINE=$pipe PTCORE_LIB=$lib timeout -k 10 120 python tools/probe_synth.py ${1:-100000} >> $OUT 2>&1 || exit 1
done
done
