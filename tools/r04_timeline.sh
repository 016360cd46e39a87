#!/bin/bash
# how far apart the waves of a trace launch retire (PTCORE_DEBUG_TIMELINE): C4, one frame at 4 and at 13 passes per frame
OUT=gpurun_out/r04_timeline.txt; : > $OUT
for mb in 163840 49152; do
  echo "== job buffers $mb MiB" >> $OUT
  PTCORE_DEBUG_TIMELINE=1 PTCORE_L_BUDGET_MB=$mb timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "ptcore timeline" | tail -${LINES_PER:-12} >> $OUT
done
cat $OUT
