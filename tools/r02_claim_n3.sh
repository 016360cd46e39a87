#!/bin/bash
OUT=gpurun_out/claim_n3.txt; : > $OUT
for rep in 1 2; do
for c in 256 512 1024; do
  echo "== PTCORE_CLAIM=$c (run $rep)" >> $OUT
  PTCORE_CLAIM=$c timeout -k 10 200 python tools/probe_synth.py 40 128 10000 100000 2>&1 | grep -v amdgpu.ids | sed -E 's/gen .* spp 16: //' >> $OUT || exit 1
done
done
