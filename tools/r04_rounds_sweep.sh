#!/bin/bash
# split rounds 0..3 with the nested-exit form behind them (rounds 0: the nested form is the whole loop), C4 and C3, one box
OUT=gpurun_out/r04/rounds_sweep.txt; mkdir -p gpurun_out/r04; : > $OUT
for cfg in "" "--config C3"; do
for r in 0 1 2 3; do
  echo "== ${cfg:-C4} PTCORE_SPLIT_ROUNDS=$r" >> $OUT
  PTCORE_SPLIT_ROUNDS=$r timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f raygen %.1f resolve %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step']))" >> $OUT || exit 1
done
done
cat $OUT
