#!/bin/bash
# PTCORE_SPLIT_ROUNDS sweep per BASELINE config on one box: tools/r03_rounds_sweep.sh C3 "1 2 3 4 6" [extra bench args]
CFG=$1; ROUNDS=${2:-"1 2 3 4"}; shift; shift
OUT=gpurun_out/r03/rounds_$CFG.txt; mkdir -p gpurun_out/r03; : > $OUT
for r in $ROUNDS; do
  echo -n "$CFG split rounds $r: " >> $OUT
  PTCORE_SPLIT_ROUNDS=$r timeout -k 10 300 python bench.py --config $CFG --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms/frame  trace %.1f glass %.1f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step']))" >> $OUT || exit 1
done
cat $OUT
