#!/bin/bash
# Thin-lens ray generation with the column walk: whole GPU suite, bench A/B, then the rocprof passes of the bench.
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/gpu_tests_s3.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r02/gpu_tests_s3.log; [ $rc -eq 0 ] || exit 1
for V in lens simple; do
  if [ $V = simple ]; then export PTCORE_RAYGEN_SIMPLE=1; else unset PTCORE_RAYGEN_SIMPLE; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_raygen_$V.json 2> gpurun_out/r02/bench_raygen_$V.err || { tail -3 gpurun_out/r02/bench_raygen_$V.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_raygen_$V.json")); f=d["roofline_fp64"]
print("raygen=$V value %.1f ms/step %.1f | trace %.1f glass %.1f raygen %.1f resolve %.1f" % (d["value"], d["ms_per_step"], f["trace_ms_per_step"], f["glass_ms_per_step"], f["raygen_ms_per_step"], f["resolve_ms_per_step"]))
PY
done
unset PTCORE_RAYGEN_SIMPLE
tools/profile.sh r02 || exit 1
