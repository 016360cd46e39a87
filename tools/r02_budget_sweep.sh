#!/bin/bash
# samples per pass against the job-buffer budget (C4): fewer, larger passes
OUT=gpurun_out/budget_sweep.txt; : > $OUT
for mb in 49152 98304 163840 24576; do
  echo "== PTCORE_L_BUDGET_MB=$mb" >> $OUT
  PTCORE_L_BUDGET_MB=$mb timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['spp_chunk'], d['roofline_fp64'])" >> $OUT || exit 1
done
