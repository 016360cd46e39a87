#!/bin/bash
# Round-2 closing run on the GPU box: the whole GPU suite, the bench, the grouped-scan probe, verification evidence.
set -o pipefail
TAG=${1:-final}
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02/gpu_tests_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r02/gpu_tests_$TAG.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_$TAG.json 2> gpurun_out/r02/bench_$TAG.err || { tail -5 gpurun_out/r02/bench_$TAG.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_$TAG.json"))
f=d["roofline_fp64"]
print("bench value %.1f ms/step %.1f | trace %.1f glass %.1f raygen %.1f resolve %.1f | fp64 frac %.3f | hbm roofline %.1f GB/s | cpu %.1f x%.0f" % (
  d["value"], d["ms_per_step"], f["trace_ms_per_step"], f["glass_ms_per_step"], f["raygen_ms_per_step"], f["resolve_ms_per_step"], f["frac"],
  d["roofline"]["achieved"], d.get("cpu_baseline",{}).get("value",0), d.get("gpu_over_cpu",0)))
PY
for R in 2 0; do
  echo "== grouped scan (33-128 objects), PTCORE_SPLIT_ROUNDS=$R" | tee -a gpurun_out/r02/wide_probe_$TAG.txt
  PTCORE_SPLIT_ROUNDS=$R timeout -k 10 300 python tools/probe_synth.py 40 64 100 128 2>&1 | sed -E 's/gen .* spp 16: //' | tee -a gpurun_out/r02/wide_probe_$TAG.txt || exit 1
done
tools/r02_evidence.sh || exit 1
