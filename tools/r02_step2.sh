#!/bin/bash
# BVH centre/half nodes: tests + probe; C4 in the wavefront form; two-rank rehearsal of bench.py over gloo; rocprof passes.
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_bvh_gpu.py tests/test_verify_modes_gpu.py tests/test_fuzz_gpu.py tests/test_wavefront_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for P in mega wavefront; do
  PTCORE_PIPELINE=$P timeout -k 10 300 python tools/probe_synth.py 10000 100000 1000000 2>&1 | sed -E 's/gen .* spp 16: //' | tee -a gpurun_out/r02/n3_probe_ch.txt || exit 1
done
PTCORE_PIPELINE=wavefront timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/r02/bench_wavefront.json 2> gpurun_out/r02/bench_wavefront.err || { tail -3 gpurun_out/r02/bench_wavefront.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r02/bench_wavefront.json')); print('C4 wavefront form: value %.1f ms/step %.1f' % (d['value'], d['ms_per_step']))"
PT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_gloo2.json 2> gpurun_out/r02/bench_gloo2.err || { tail -5 gpurun_out/r02/bench_gloo2.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r02/bench_gloo2.json').read().strip().splitlines()[-1]); print('2-rank gloo rehearsal: value %.1f ms/step %.1f per-rank %s' % (d['value'], d['ms_per_step'], d['per_rank_render_ms_per_step']))"
