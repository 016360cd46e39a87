#!/bin/bash
# Synthetic BVH scenes: the yield threshold of the node walk (PTCORE_BVH_NODE_MIN; 0 = walk until every lane has objects, the
# round-1 behaviour) in both forms of the loop.  Usage: tools/r02_n3_sweep.sh <tag>; SIZES="10000 100000" by default.
set -o pipefail
TAG=${1:-a}
mkdir -p gpurun_out/r02
for P in mega wavefront; do
  for NM in ${NODE_MINS:-0 16 32 48}; do
    echo "== PTCORE_PIPELINE=$P PTCORE_BVH_NODE_MIN=$NM" | tee -a gpurun_out/r02/n3_sweep_$TAG.txt
    PTCORE_PIPELINE=$P PTCORE_BVH_NODE_MIN=$NM timeout -k 10 300 python tools/probe_synth.py ${SIZES:-10000 100000} 2>&1 | tee -a gpurun_out/r02/n3_sweep_$TAG.txt || exit 1
  done
done
