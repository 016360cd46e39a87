#!/bin/bash
# what the wave-cooperative primary pass does and costs (PTCORE_DEBUG_PASS_LOG), and the counters rocprofv3 offers for vL1D / TA
OUT=gpurun_out/r04/n3_passlog.txt; mkdir -p gpurun_out/r04; : > $OUT
for n in 10000 100000 1000000; do
  echo "== n=$n" >> $OUT
  PTCORE_DEBUG_PASS_LOG=1 timeout -k 10 300 python tools/probe_synth.py $n >> $OUT 2>&1 || exit 1
  PTCORE_PRIMARY=lane PTCORE_DEBUG_PASS_LOG=1 timeout -k 10 300 python tools/probe_synth.py $n >> $OUT 2>&1 || exit 1
done
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TA\|TD\|TCC\)_[A-Z0-9_a-z]*" | sort -u > gpurun_out/r04/counters_avail.txt
cat $OUT
wc -l gpurun_out/r04/counters_avail.txt
