#!/bin/bash
# Round-4 closing runs on the GPU box (two gpurun calls: `tools/r04_final.sh bench <tag>` and `tools/r04_final.sh prof <tag>`):
#   bench: the whole GPU suite, smoke(), the default bench line and one line per other BASELINE config
#   prof : rocprofv3 kernel stats + PMC passes of the same commands (tools/profile.sh), C4 / C3 / C2 / C5
set -o pipefail
WHAT=${1:-bench}; TAG=${2:-r04f}
O=gpurun_out/r04f; mkdir -p $O
if [ "$WHAT" = bench ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests_$TAG.log 2>&1
  rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests_$TAG.log; [ $rc -eq 0 ] || exit 1
  timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
  timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
  for C in C2 C3 C5; do
    c=$(echo $C | tr C c)
    case $C in C2) ST="--steps 20 --warmup 3";; C3) ST="--steps 5 --warmup 1";; C5) ST="--steps 1 --warmup 1";; esac
    timeout -k 10 400 python bench.py --config $C $ST --no-cpu-baseline > $O/${TAG}_${c}_bench.json 2> $O/${TAG}_${c}_bench.err || { tail -5 $O/${TAG}_${c}_bench.err; exit 1; }
  done
  python - <<PY
import json
for c in ("", "_c2", "_c3", "_c5"):
    d = json.loads(open("$O/${TAG}%s_bench.json" % c).read().strip().splitlines()[-1])
    f = d["roofline_fp64"]
    print("%-4s %9.1f Mseg/s %9.2f ms | trace %.1f glass %.1f raygen %.1f resolve %.1f | fp64 frac %.3f (%.3f at %s MHz) | hbm %.0f GB/s | cpu %s" % (
        c or "_c4", d["value"], d["ms_per_step"], f["trace_ms_per_step"], f["glass_ms_per_step"], f["raygen_ms_per_step"], f["resolve_ms_per_step"],
        f["frac"], f.get("frac_at_measured_clock", 0), f.get("shader_clock_mhz"), d["roofline"]["achieved"], d.get("cpu_baseline", {}).get("value")))
PY
else
  tools/profile.sh $TAG || exit 1
  tools/profile.sh ${TAG}_c3 --config C3 || exit 1
  tools/profile.sh ${TAG}_c2 --config C2 || exit 1
  tools/profile.sh ${TAG}_c5 --config C5 || exit 1
fi
