#!/bin/bash
mkdir -p gpurun_out/r03; OUT=gpurun_out/r03/bvh_knob_sweep.txt; : > $OUT
for nm in 8 12 16 24 32; do for ml in 16 24 32; do
  echo "node_min $nm min_lanes $ml: $(PTCORE_BVH_NODE_MIN=$nm PTCORE_BVH_MIN_LANES=$ml timeout -k 10 100 python tools/probe_synth.py 100000 2>&1 | grep -o 'scan [0-9.]*ms')" >> $OUT
done; done
cat $OUT
