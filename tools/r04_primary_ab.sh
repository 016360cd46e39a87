#!/bin/bash
# BVH path: the wave-cooperative primary pass (default) against the round-3 loop (PTCORE_PRIMARY=lane), one box; then verify_bvh
OUT=gpurun_out/r04/primary_ab.txt; mkdir -p gpurun_out/r04; : > $OUT
for n in ${SIZES:-10000 100000 1000000}; do
  for mode in coop lane coop lane; do
    echo "== n=$n PTCORE_PRIMARY=$mode" >> $OUT
    PTCORE_PRIMARY=$mode PTCORE_VERBOSE=${VERBOSE:-} timeout -k 10 300 python tools/probe_synth.py $n >> $OUT 2>&1 || exit 1
  done
done
echo "== verify_bvh, 20000 objects" >> $OUT
PTCORE_SCAN=verify_bvh timeout -k 10 300 python tools/probe_synth.py 20000 >> $OUT 2>&1 || exit 1
cat $OUT
