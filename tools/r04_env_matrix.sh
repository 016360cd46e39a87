#!/bin/bash
# Parity tests of the final binary under the library's A/B knobs: tools/r04_env_matrix.sh [knob=value ...]   (default: the whole matrix)
# Output: gpurun_out/r04m/env_matrix.txt (last line of pytest per knob set, plus the failures' short summary).
mkdir -p gpurun_out/r04m
OUT=gpurun_out/r04m/env_matrix.txt; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_split_passes_gpu.py tests/test_edge_scenes_gpu.py tests/test_grazing_gpu.py -q -m gpu -rf 2>&1 | grep -E "^FAILED|^E  |passed|failed" | cut -c1-300 | head -12 >> $OUT; }
if [ $# -gt 0 ]; then for k in "$@"; do run $k; done; cat $OUT; exit 0; fi
run PTCORE_SPLIT_ROUNDS=0
run PTCORE_TAIL=trip
run PTCORE_SPLIT_ROUNDS=0 PTCORE_TAIL=trip
run PTCORE_SCAN=bvh PTCORE_PRIMARY=lane
run PTCORE_GATHER=rccl
run PTCORE_AUTO_GROW=0
run PTCORE_SPLIT_ROUNDS=1
run PTCORE_SPLIT_ROUNDS=5
run PTCORE_BLOCKS_PER_CU=3
run PTCORE_L_BUDGET_MB=64
run PTCORE_CLAIM=64
run PTCORE_PIPELINE=wavefront
run PTCORE_SCAN=uniform
run PTCORE_SCAN=wide
run PTCORE_SCAN=bvh
run PTCORE_SCAN=bvh PTCORE_PIPELINE=walk32
run PTCORE_SCAN=verify
cat $OUT
