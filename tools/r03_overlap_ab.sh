#!/bin/bash
# Same-box A/B: ray generation of the next chunk beside the trace passes of the running one (PTCORE_RAYGEN_OVERLAP) on bench.py's C4 frame
OUT=gpurun_out/r03/overlap_ab.txt; mkdir -p gpurun_out/r03; : > $OUT
run() {
  echo "== overlap $1 budget $2 MiB" >> $OUT
  PTCORE_RAYGEN_OVERLAP=$1 PTCORE_L_BUDGET_MB=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms/frame  chunk %d  trace %.1f glass %.1f raygen %.1f resolve %.1f' % (d['value'], d['ms_per_step'], d['config']['spp_chunk'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step']))" >> $OUT || exit 1
}
for rep in 1 2; do run 0 163840; run 1 163840; run 1 196608; run 1 49152; run 0 49152; done
cat $OUT
