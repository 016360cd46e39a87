#!/bin/bash
# rocprofv3 of the BVH probe (tools/probe_synth.py) for one pipeline: kernel stats + one SQ counter pass.
#   tools/r03_prof_walk.sh <tag> <pipeline> [objects] [seed]      -> gpurun_out/r03/prof_<tag>/{stats.csv,pmc.csv,probe.txt}
TAG=$1; export PTCORE_PIPELINE=${2:-walk32}; N=${3:-100000}; export PROBE_SEED=${4:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03/prof_$TAG; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_synth.py $N > $OUT/probe.txt 2> $OUT/trace.err || { echo "trace pass failed"; tail -3 $OUT/trace.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); test -n "$f" && cp "$f" $OUT/stats.csv
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -- python3 tools/probe_synth.py $N > $OUT/pmc.txt 2> $OUT/pmc.err || { echo "pmc pass failed"; tail -3 $OUT/pmc.err; exit 1; }
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1); test -n "$f" && python3 tools/r03_pmc_sum.py "$f" > $OUT/pmc_by_kernel.txt
rm -rf $OUT/trace $OUT/pmc
grep "n=" $OUT/probe.txt | sed -E "s/gen .* spp 16: //"; test -f $OUT/stats.csv && cut -d, -f1-4 $OUT/stats.csv | head -9; cat $OUT/pmc_by_kernel.txt
