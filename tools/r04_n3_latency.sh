#!/bin/bash
# vL1D latency of the BVH walk under load (one more --pmc pass next to tools/r04_profile_n3.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04_n3lat; mkdir -p $OUT
for PMC in "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_BUSY_max TA_BUSY_min" "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  t=$(echo $PMC | cut -d' ' -f1)
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/$t -- python3 tools/probe_synth.py 100000 > $OUT/$t.txt 2> $OUT/$t.err || { echo "pass $t failed"; tail -3 $OUT/$t.err; }
done
python3 - <<'PY'
import csv, glob
tot = {}
for f in glob.glob("gpurun_out/prof_r04_n3lat/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
