#!/bin/bash
# same-box sweep of environment knobs on the headline bench: tools/r02_env_sweep.sh "VAR=a" "VAR=b" ...
OUT=gpurun_out/env_sweep.txt; : > $OUT
for rep in 1 2; do
for kv in "$@"; do
  echo "== $kv (run $rep)" >> $OUT
  env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f raygen %.1f resolve %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step']))" >> $OUT || exit 1
done
done
