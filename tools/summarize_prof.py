#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*.{csv,json}."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag}
for f in glob.glob(src + "/trace/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("profiles/%s_kernel_stats.csv" % tag, "w") as g:
        g.write(open(f).read())
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")}
                           for r in rows[:5]]
try:
    out["bench_under_trace"] = json.load(open(src + "/bench_under_trace.json"))
except Exception as e:  # noqa: BLE001
    out["bench_under_trace"] = str(e)
pmc = {}
for f in sorted(glob.glob(src + "/pmc*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "trace_kernel" in name:
            key = "trace_kernel"
        elif "resolve_kernel" in name:
            key = "resolve_kernel"
        else:
            continue
        pmc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {}
for k, d in pmc.items():
    summary[k] = {c: {"per_launch": v} for c, v in d.items()}
out["pmc"] = summary
t = pmc.get("trace_kernel", {})


def g(c):
    v = t.get(c)
    return v[0] if v else 0.0  # launch 0 is a full spp chunk, the same shape as every launch of the full run


jobs = None
try:
    import re
    b3 = json.load(open(src + "/pmc3.json"))
    m = re.search(r"(\d+)x(\d+), (\d+) spp", b3["config"]["workload"])
    jobs = int(m.group(1)) * int(m.group(2)) * min(int(b3["config"]["spp_chunk"]), int(m.group(3)))
except Exception as e:  # noqa: BLE001
    out["jobs_error"] = str(e)
if t:
    fetch_b = g("FETCH_SIZE") * 1024.0 * 2.0  # KB; x2: gfx950 FETCH_SIZE reports half of a wide coalesced read (guide, HBM section)
    write_b = g("WRITE_SIZE") * 1024.0
    out["derived_trace_kernel"] = {
        "waves": g("SQ_WAVES"),
        "valu_insts_per_wave": g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1),
        "valu_lane_utilisation": (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
                                  if g("SQ_ACTIVE_INST_VALU") else None),
        "valu_busy_share_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1),
        "f64_add": g("SQ_INSTS_VALU_ADD_F64"), "f64_mul": g("SQ_INSTS_VALU_MUL_F64"),
        "f64_fma": g("SQ_INSTS_VALU_FMA_F64"), "f64_trans": g("SQ_INSTS_VALU_TRANS_F64"),
        "int32": g("SQ_INSTS_VALU_INT32"), "int64": g("SQ_INSTS_VALU_INT64"), "cvt": g("SQ_INSTS_VALU_CVT"),
        "valu_total": g("SQ_INSTS_VALU"), "salu": g("SQ_INSTS_SALU"), "smem": g("SQ_INSTS_SMEM"),
        "hbm_fetch_bytes_x2": fetch_b, "hbm_write_bytes": write_b,
    }
    json.dump({"trace_kernel": {"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_corrected_x2": fetch_b,
                                "write_bytes": write_b, "samples_in_measured_launch": jobs,
                                "hbm_bytes_per_sample": (fetch_b + write_b) / jobs if jobs else None,
                                "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                          "separate passes; FETCH_SIZE doubled per the gfx950 correction)" % tag}},
              open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
print(json.dumps(out.get("derived_trace_kernel"), indent=1))
print(json.dumps(out.get("kernel_stats"), indent=1))
