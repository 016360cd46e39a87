#!/bin/bash
# same-box A/B of the thin-lens ray generation: round 2's walk down a lane's column (PTCORE_RAYGEN=column) against the wave's pool of jobs
OUT=gpurun_out/r04_raygen_pool_ab.txt; : > $OUT
for rep in 1 2; do
for form in column pool; do
  echo "== PTCORE_RAYGEN=$form (run $rep)" >> $OUT
  PTCORE_RAYGEN=$form PTCORE_L_BUDGET_MB=163840 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f raygen %.1f resolve %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step']))" >> $OUT || exit 1
done
done
cat $OUT
