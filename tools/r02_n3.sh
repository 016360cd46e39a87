#!/bin/bash
# BVH scenes: GPU tests in the wavefront form (the default for BVH scenes), then the synthetic probe in both forms.
# Stops at the first failing step: no GPU step is started after a fault.
set -o pipefail
TAG=${1:-a}
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_bvh_gpu.py tests/test_fuzz_gpu.py tests/test_edge_scenes_gpu.py tests/test_verify_modes_gpu.py -x -q -m gpu > gpurun_out/r02/n3_tests_$TAG.log 2>&1
rc=$?; echo "pytest (wavefront default) rc=$rc"; tail -4 gpurun_out/r02/n3_tests_$TAG.log; [ $rc -eq 0 ] || exit 1
PTCORE_PIPELINE=wavefront timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_edge_scenes_gpu.py -x -q -m gpu > gpurun_out/r02/n3_tests_flat_$TAG.log 2>&1
rc=$?; echo "pytest (wavefront forced on flat scans) rc=$rc"; tail -4 gpurun_out/r02/n3_tests_flat_$TAG.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
for P in ${PIPES:-wavefront mega}; do
  PTCORE_PIPELINE=$P timeout -k 10 400 python tools/probe_synth.py ${SIZES:-10000 100000 1000000} 2>&1 | tee -a gpurun_out/r02/n3_probe_$TAG.txt || exit 1
done
