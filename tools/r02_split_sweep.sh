#!/bin/bash
# GPU suite, then the bench with 2 / 0 / 1 / 3 split rounds (run on the GPU box through gpurun).
set -o pipefail
mkdir -p gpurun_out/r02
TAG=${1:-a}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/gpu_tests_$TAG.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r02/gpu_tests_$TAG.log
for R in ${ROUNDS:-2 0 1 3}; do
  PTCORE_SPLIT_ROUNDS=$R timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/r02/bench_${TAG}_split$R.json 2> gpurun_out/r02/bench_${TAG}_split$R.err || { echo "bench R=$R failed"; tail -5 gpurun_out/r02/bench_${TAG}_split$R.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_${TAG}_split$R.json"))
print("R=$R value %.1f ms/step %.1f trace avg launch %.2f ms fp64 frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_fp64"]["frac"]))
PY
done
