#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun).
#   tools/profile.sh <tag> [bench args...]
# Pass 0: --kernel-trace --stats of the full default bench command.
# PMC passes (each its own run, counters only, as the pool requires) use a short
# workload with the SAME launch shape as the full run (2 trace launches, the first a full chunk).
set -o pipefail
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
SHORT="--spp ${SHORT_SPP:-192} --steps 1 --warmup 0 --no-cpu-baseline"   # the launch shapes of the full run at a fraction of its samples
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
echo "trace pass done"
i=0
for PMC in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py $SHORT "$@" > $OUT/pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; exit 1; }
  echo "pmc pass $i done"
  i=$((i+1))
done
