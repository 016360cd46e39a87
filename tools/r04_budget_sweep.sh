#!/bin/bash
# job-buffer budget against steady frame time and first-frame (allocation) time, C4 through pt_render; claim size fixed or by default
OUT=gpurun_out/r04/budget_sweep.txt; mkdir -p gpurun_out/r04; : > $OUT
for claim in default 256 512; do
for mb in 8192 16384 32768 49152 98304 163840; do
  if [ $claim = default ]; then unset PTCORE_CLAIM; else export PTCORE_CLAIM=$claim; fi
  echo -n "claim $claim budget $mb MiB: " >> $OUT
  FRAMES=3 PTCORE_L_BUDGET_MB=$mb timeout -k 10 300 python tools/r04_pt_render_default.py --keep-env | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('passes %d  steady %.1f ms  first frame %.0f ms (+%.0f over steady)  %.0f Mseg/s' % (d['passes_per_frame'], d['steady_frame_ms_median'], d['first_frame_ms'], d['first_frame_ms']-d['steady_frame_ms_median'], d['msegments_per_s_steady']))" >> $OUT || exit 1
done
done
cat $OUT
