#!/bin/bash
# same box, same binary: the pass behind the split rounds with the exit search nested (default) against the round-1 form
# (PTCORE_TAIL=trip: the exit search is the lane's next trip), on C4, C3, C5 (C5 at 512 spp) and C2
OUT=gpurun_out/r04/tail_ab.txt; mkdir -p gpurun_out/r04; : > $OUT
run() {  # label, env, bench args
  echo "== $1" >> $OUT
  env $2 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['roofline_fp64']; print('%.1f Mseg/s  %.2f ms  trace %.1f glass %.1f raygen %.1f resolve %.1f  differing bytes %s' % (d['value'], d['ms_per_step'], f['trace_ms_per_step'], f['glass_ms_per_step'], f['raygen_ms_per_step'], f['resolve_ms_per_step'], d.get('rgba8_bytes_differing')))" >> $OUT || exit 1
}
for rep in 1 2; do
  run "C4 nested (run $rep)" "PTCORE_TAIL=nested" ""
  run "C4 trip   (run $rep)" "PTCORE_TAIL=trip" ""
done
run "C3 nested" "PTCORE_TAIL=nested" "--config C3"
run "C3 trip" "PTCORE_TAIL=trip" "--config C3"
run "C2 nested" "PTCORE_TAIL=nested" "--config C2"
run "C2 trip" "PTCORE_TAIL=trip" "--config C2"
cat $OUT
