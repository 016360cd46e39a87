#!/bin/bash
# Synthetic BVH scenes: leaf passes (all objects of a node at once / one per pass) x straggler threshold, both forms.
set -o pipefail
TAG=${1:-a}
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_bvh_gpu.py tests/test_verify_modes_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
PTCORE_BVH_LEAF_SINGLE=1 timeout -k 10 300 python -m pytest tests/test_bvh_gpu.py tests/test_verify_modes_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
PTCORE_BVH_LEAF_SINGLE=1 PTCORE_PIPELINE=mega timeout -k 10 300 python -m pytest tests/test_bvh_gpu.py tests/test_verify_modes_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for P in mega wavefront; do
  for CFG in "0 24" "1 24" "1 12" "1 40" "0 12"; do
    set -- $CFG
    echo "== PTCORE_PIPELINE=$P PTCORE_BVH_LEAF_SINGLE=$1 PTCORE_BVH_MIN_LANES=$2 (node_min 16)" | tee -a gpurun_out/r02/n3_sweep2_$TAG.txt
    PTCORE_PIPELINE=$P PTCORE_BVH_LEAF_SINGLE=$1 PTCORE_BVH_MIN_LANES=$2 PTCORE_WF_MIN_LANES=$2 timeout -k 10 300 python tools/probe_synth.py ${SIZES:-10000 100000 1000000} 2>&1 | sed -E 's/gen .* spp 16: //' | tee -a gpurun_out/r02/n3_sweep2_$TAG.txt || exit 1
  done
done
