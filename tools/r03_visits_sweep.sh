#!/bin/bash
# Sweep of the all-in-one BVH loop's trip policy (seed 2: no fat ray): PTCORE_BVH_MAX_VISITS x PTCORE_BVH_MIN_LANES at <objects>
OUT=gpurun_out/r03/visits_sweep_$1.txt; mkdir -p gpurun_out/r03; : > $OUT
for mv in 0 8 12 16 24 32; do for ml in ${MLS:-24}; do
  echo -n "max_visits $mv min_lanes $ml: " >> $OUT
  PROBE_SEED=2 PTCORE_BVH_MAX_VISITS=$mv PTCORE_BVH_MIN_LANES=$ml timeout -k 10 120 python tools/probe_synth.py $1 2>&1 | grep "n=" | sed -E "s/.*scan ([0-9.]+)ms.*kernel rate ([0-9.]+).*/\1 ms  \2 Mseg\/s/" >> $OUT || exit 1
done; done
cat $OUT
