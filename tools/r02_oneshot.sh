#!/bin/bash
# one-shot CLI renders of C4 (process start to PNG on disk) against the job-buffer budget
OUT=gpurun_out/oneshot.txt; : > $OUT
for mb in 49152 8192 4096 16384; do
  echo "== PTCORE_L_BUDGET_MB=$mb" >> $OUT
  ( time PTCORE_L_BUDGET_MB=$mb timeout -k 10 120 path_trace_golang_amd/render -headless -gpu -scene scenes/gpu_showcase.json -width 1920 -height 1080 -spp 1024 -depth 8 -out gpurun_out/oneshot.png ) >> $OUT 2>&1 || exit 1
done
