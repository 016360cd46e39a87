#!/bin/bash
# same-box A/B of builds of libptcore.so on the BVH probe: OBJS="10000 100000" tools/ab_probe.sh <libA> <libB> [<libC> ...]
OUT=gpurun_out/ab_probe.txt; : > $OUT
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (run $rep)" >> $OUT
  PTCORE_LIB=$PWD/$lib timeout -k 10 200 python tools/probe_synth.py ${OBJS:-100000} 2>&1 | grep -v amdgpu.ids | sed -E 's/gen .* spp 16: //' >> $OUT || exit 1
done
done
