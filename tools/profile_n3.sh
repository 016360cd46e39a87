#!/bin/bash
# rocprofv3 passes for the BVH path on a synthetic scene (run on the GPU box through gpurun).
#   tools/profile_n3.sh <tag> [objects] [mega|wavefront]
set -o pipefail
TAG=${1:-r02_n3}; N=${2:-100000}; export PTCORE_PIPELINE=${3:-mega}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_synth.py $N > $OUT/probe.txt 2> $OUT/trace.err || exit 1
echo "trace pass done"
i=0
for PMC in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ; do
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 tools/probe_synth.py $N > $OUT/pmc$i.txt 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; exit 1; }
  echo "pmc pass $i done"
  i=$((i+1))
done
