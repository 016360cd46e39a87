#!/bin/bash
# PTCORE_REFILL_MIN sweep (a wave refills only once that many lanes are idle), C3 and C4, one box
OUT=gpurun_out/r04_refill_min.txt; : > $OUT
for cfg in "--config C3" ""; do
for m in 1 2 4 8 16 32; do
  echo -n "${cfg:-C4} PTCORE_REFILL_MIN=$m: " >> $OUT
  PTCORE_REFILL_MIN=$m timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step']))" >> $OUT || exit 1
done
done
cat $OUT
