#!/bin/bash
# FETCH_SIZE and the L2's memory-side request counters by size (32 / 64 / 128 B) for (1) independent random 128-byte records of a 2 GiB
# table read with 7 or 3 sixteen-byte loads per lane (known count) and (2) the BVH probe at 10^5 objects.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04_fetch; mkdir -p $OUT
export NODE_FETCH_SCATTER=1
i=0
for PMC in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_READ_sum TCC_READ_SECTORS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/control$i -- tools/node_fetch_bench > $OUT/control$i.json 2> $OUT/control$i.err || { echo "control pass $i ($PMC) failed"; tail -3 $OUT/control$i.err; }
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/probe$i -- python3 tools/probe_synth.py 100000 > $OUT/probe$i.txt 2> $OUT/probe$i.err || { echo "probe pass $i ($PMC) failed"; tail -3 $OUT/probe$i.err; }
  echo "pass $i done"
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, json
out = {}
src = "gpurun_out/prof_r04_fetch"
bench = json.load(open(src + "/control0.json"))["results"]
sc = [b for b in bench if b.get("scatter")]
ctl = {}
for f in sorted(glob.glob(src + "/control*/**/*_counter_collection.csv", recursive=True)):
    disp = {}
    for r in csv.DictReader(open(f)):
        if "true" not in r["Kernel_Name"]:
            continue
        disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    for k, b in enumerate(sc):  # two launches per configuration, the second one timed
        if len(ids) >= 2 * k + 2:
            ctl.setdefault("%d loads" % b["loads_per_visit"], dict(b)).update(disp[ids[2 * k + 1]])
for k, e in ctl.items():
    v = e["record_visits"]
    e["per_visit"] = {c: e[c] / v for c in e if c.isupper() or c.startswith("TCC") or c == "FETCH_SIZE"}
    if "FETCH_SIZE" in e:
        e["per_visit"]["FETCH_SIZE_bytes"] = e["FETCH_SIZE"] * 1024.0 / v
out["independent_random_records_2GiB"] = ctl
probe = {}
for f in sorted(glob.glob(src + "/probe*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] or "primary_bvh" in r["Kernel_Name"]:
            name = "trace_kernel" if "trace_kernel" in r["Kernel_Name"] else "primary_bvh_kernel"
            probe.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
            probe[name][r["Counter_Name"]] += float(r["Counter_Value"])
out["bvh_probe_1e5_objects_two_frames"] = probe
json.dump(out, open("gpurun_out/r04/r04_fetch_control.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
