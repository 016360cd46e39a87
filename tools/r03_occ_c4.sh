#!/bin/bash
# Same-box A/B of the headline kernel's occupancy on bench.py's C4 frame:
#   5 waves/SIMD (shipping: launch bounds 5, 92 VGPRs), 4 waves/SIMD (same binary, PTCORE_BLOCKS_PER_CU=4),
#   6 waves/SIMD (-DPT_FLAT_WAVES=6 build: 80 VGPRs, 17 spilled in the split kernel).
# usage: tools/r03_occ_c4.sh <lib_w5> <lib_w6>      -> gpurun_out/r03/occ_c4.txt
OUT=gpurun_out/r03/occ_c4.txt; mkdir -p gpurun_out/r03; : > $OUT
run() {  # label, lib, blocks-per-cu
  echo "== $1" >> $OUT
  PTCORE_BLOCKS_PER_CU=$3 PTCORE_LIB=$PWD/$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms/frame  trace %.1f glass %.1f raygen %.1f resolve %.1f  fp64 frac %.4f' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step'], f['frac']))" >> $OUT || exit 1
}
for rep in 1 2; do
  run "5 waves/SIMD (shipping binary), run $rep" $1 8 || exit 1
  run "4 waves/SIMD (shipping binary, PTCORE_BLOCKS_PER_CU=4), run $rep" $1 4 || exit 1
  run "6 waves/SIMD (-DPT_FLAT_WAVES=6 build), run $rep" $2 8 || exit 1
  run "3 waves/SIMD (shipping binary, PTCORE_BLOCKS_PER_CU=3), run $rep" $1 3 || exit 1
done
cat $OUT
