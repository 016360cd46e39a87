#!/bin/bash
# the drop-in's default path next to the headline, one box: pt_render with library defaults (twice: fresh process each), the same with the
# bench's job-buffer budget, and bench.py itself
mkdir -p gpurun_out/r04
python tools/r04_pt_render_default.py > gpurun_out/r04/pt_render_default.json || exit 1
python tools/r04_pt_render_default.py > gpurun_out/r04/pt_render_default_b.json || exit 1
PTCORE_L_BUDGET_MB=163840 python tools/r04_pt_render_default.py --keep-env > gpurun_out/r04/pt_render_budget160.json || exit 1
python bench.py --no-cpu-baseline > gpurun_out/r04/bench_same_box.json 2>/dev/null || exit 1
cat gpurun_out/r04/pt_render_default.json gpurun_out/r04/pt_render_default_b.json gpurun_out/r04/pt_render_budget160.json
python -c "
import json; d=json.load(open('gpurun_out/r04/bench_same_box.json')); print('bench.py: %.1f Mseg/s %.2f ms' % (d['value'], d['ms_per_step']))"
