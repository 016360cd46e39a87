#!/bin/bash
# same-box A/B of builds of libptcore.so on the headline bench: tools/ab_bench.sh <libA> <libB> ...   (BUDGETS="163840 49152")
OUT=gpurun_out/ab_bench.txt; : > $OUT
for rep in 1 2; do
for mb in ${BUDGETS:-163840}; do
for lib in "$@"; do
  echo "== $lib budget $mb MiB (run $rep)" >> $OUT
  PTCORE_L_BUDGET_MB=$mb PTCORE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f raygen %.1f resolve %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step']))" >> $OUT || exit 1
done
done
done
