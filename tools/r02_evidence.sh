#!/bin/bash
# Round-2 correctness evidence on the GPU box (-> profiles/r02_scan_verification.txt):
#   every scan of the shipping kernels (split trace passes + glass_kernel, grouped scan, BVH walk in both forms of the
#   loop) is repeated by the plain object-by-object loop of the reference and disagreements are counted.
set -o pipefail
OUT=gpurun_out/r02/scan_verification.txt
mkdir -p gpurun_out/r02
: > $OUT
echo "PTCORE_SCAN=verify on BASELINE configs C2-C5 at their FULL sizes, seed 1 (tools/verify_full_configs.py): split trace passes + glass_kernel + all-in-one pass, round-2 kernels" >> $OUT
timeout -k 10 600 python tools/verify_full_configs.py 2>&1 | tee -a $OUT || exit 1
echo >> $OUT
echo "PTCORE_SCAN=verify_wide on synthetic 40 / 64 / 100 / 128-object scenes at 1920x1080 (tools/verify_wide.py): grouped scan, split passes" >> $OUT
VERIFY_SPP=${VERIFY_SPP:-24} timeout -k 10 600 python tools/verify_wide.py 2>&1 | tee -a $OUT || exit 1
echo >> $OUT
for P in mega wavefront; do
  echo "PTCORE_SCAN=verify_bvh, PTCORE_PIPELINE=$P on synthetic scenes (tools/probe_synth.py 300 3000; the verify build runs the every-object scan too, hence the rates)" >> $OUT
  PTCORE_SCAN=verify_bvh PTCORE_PIPELINE=$P timeout -k 10 600 python tools/probe_synth.py 300 3000 2>&1 | sed -E 's/gen .* spp 16: //' | tee -a $OUT || exit 1
  echo "PTCORE_SCAN=verify_bvh, PTCORE_PIPELINE=$P on very large synthetic scenes, small frames (tools/verify_big_bvh.py)" >> $OUT
  PTCORE_PIPELINE=$P timeout -k 10 600 python tools/verify_big_bvh.py 2>&1 | tee -a $OUT || exit 1
  echo >> $OUT
done
echo "PTCORE_SCAN=verify_bvh, ray binning on (PTCORE_WF_SORT=1), wavefront form" >> $OUT
PTCORE_SCAN=verify_bvh PTCORE_WF_SORT=1 PTCORE_PIPELINE=wavefront timeout -k 10 600 python tools/probe_synth.py 300 3000 2>&1 | sed -E 's/gen .* spp 16: //' | tee -a $OUT || exit 1
