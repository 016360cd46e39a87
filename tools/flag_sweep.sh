#!/bin/bash
# Compiler-flag A/B on the GPU box: every build/variants/*.so (made with extra hipcc flags) takes the place of
# libptcore.so in this scratch copy for one short bench run.   tools/flag_sweep.sh
cd "$GRAFT_REPO_ROOT"
cp path_trace_golang_amd/libptcore.so /tmp/libptcore_orig.so
for v in build/variants/*.so; do
  cp "$v" path_trace_golang_amd/libptcore.so
  for rep in 1 2; do
    python3 bench.py --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
cp /tmp/libptcore_orig.so path_trace_golang_amd/libptcore.so
