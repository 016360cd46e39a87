#!/bin/bash
# Round-4 correctness evidence of the final binary on the GPU box (-> profiles/${TAG:-r04}_scan_verification.txt, ${TAG:-r04}_fuzz_soak.txt):
# every scan of the shipping kernels is repeated by the plain object-by-object loop of the reference and disagreements are counted;
# then the fuzz soak (random scenes, every closest-hit strategy against the oracle).
set -o pipefail
O=gpurun_out/r04v; mkdir -p $O
OUT=$O/${TAG:-r04}_scan_verification.txt; : > $OUT
echo "PTCORE_SCAN=verify on BASELINE configs C2-C5 at their FULL sizes, seed 1 (tools/verify_full_configs.py): split trace passes + glass_kernel + the nested-exit pass behind them, final round-4 binary" >> $OUT
timeout -k 10 600 python tools/verify_full_configs.py 2>&1 | tee -a $OUT || exit 1
echo >> $OUT
echo "PTCORE_SCAN=verify_wide on synthetic 40 / 64 / 100 / 128-object scenes at 1920x1080 (tools/verify_wide.py): grouped scan, split passes" >> $OUT
VERIFY_SPP=24 timeout -k 10 600 python tools/verify_wide.py 2>&1 | tee -a $OUT || exit 1
echo >> $OUT
echo "PTCORE_SCAN=verify_bvh on 10^5 / 10^6 objects (tools/verify_big_bvh.py): wave-cooperative primary pass (its own check against the plain loop) + all-in-one loop" >> $OUT
timeout -k 10 600 python tools/verify_big_bvh.py 2>&1 | tee -a $OUT || exit 1
S=$O/${TAG:-r04}_fuzz_soak.txt
echo "PT_SOAK_SECONDS=${PT_SOAK_SECONDS:-300} python -m pytest tests/test_fuzz_gpu.py -s -k soak  (one MI355X, final round-4 binary; six strategies: broad, wide, bvh, uniform, bvh/wavefront, bvh/walk32)" > $S
PT_SOAK_SECONDS=${PT_SOAK_SECONDS:-300} timeout -k 10 900 python -m pytest tests/test_fuzz_gpu.py -s -q -k soak >> $S 2>&1 || { tail -5 $S; exit 1; }
tail -2 $S
