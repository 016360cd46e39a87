#!/bin/bash
# Round 4, VERDICT item 2(b): rocprofv3 passes of the BVH path with the vector-L1 / address-unit counters, for both forms of the
# primary segment, and a control for FETCH_SIZE on divergent 16-byte requests (tools/node_fetch_bench: every lane chases its own
# chain of 128-byte records, 7 or 3 sixteen-byte loads per record, tables from 1 MiB to 512 MiB).
#   tools/r04_profile_n3.sh <tag> [objects]
set -o pipefail
TAG=${1:-r04_n3}; N=${2:-100000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
for MODE in coop lane; do
  export PTCORE_PRIMARY=$MODE
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$MODE/trace -- python3 tools/probe_synth.py $N > $OUT/$MODE.probe.txt 2> $OUT/$MODE.trace.err || exit 1
  echo "$MODE: trace pass done"
  i=0
  for PMC in \
    "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
    "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
    "FETCH_SIZE" \
    "WRITE_SIZE" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
    "TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
    "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
    "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
    "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
    "TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum" \
    "GRBM_GUI_ACTIVE" ; do
    rocprofv3 --pmc $PMC --output-format csv -d $OUT/$MODE/pmc$i -- python3 tools/probe_synth.py $N > $OUT/$MODE.pmc$i.txt 2> $OUT/$MODE.pmc$i.err || { echo "$MODE pmc pass $i ($PMC) failed"; tail -3 $OUT/$MODE.pmc$i.err; }
    echo "$MODE: pmc pass $i done"
    i=$((i+1))
  done
done
unset PTCORE_PRIMARY
# control: FETCH_SIZE / TCP counters of the pointer chase with known record counts
hipcc --offload-arch=gfx950 -O3 tools/node_fetch_bench.hip -o tools/node_fetch_bench 2>/dev/null
for PMC in "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $PMC | cut -d' ' -f1)
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/control/$tag -- tools/node_fetch_bench > $OUT/control.$tag.json 2> $OUT/control.$tag.err || echo "control pass $tag failed"
  echo "control: $tag done"
done
