#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*.{csv,json}.

Kernels are grouped by what they are: the split form of the trace kernel ("trace_split", the dominant launches),
its all-in-one form ("trace_allinone"), glass_kernel, raygen, resolve.  PMC values are summed over all launches
of a class in the profiled command; HBM bytes = FETCH_SIZE x 2 (gfx950: the counter reports half of a wide
coalesced read, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB units."""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag}


def klass(name: str):
    m = re.search(r"trace_kernel<(\w+), (\w+), (\d+), (\w+)>", name)
    if m:
        # 4th parameter: FORM (0 all-in-one, 1 split, 2 nested exit search); a bool before round 4
        return "trace_split" if m.group(4) in ("true", "1") else "trace_nested" if m.group(4) == "2" else "trace_allinone"
    for k in ("glass_kernel", "raygen_lens_pool_kernel", "raygen_lens_kernel", "raygen_kernel", "resolve_kernel", "wf_", "untile_kernel"):
        if k in name:
            return k.replace("_kernel", "")
    return None


for f in glob.glob(src + "/trace/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("profiles/%s_kernel_stats.csv" % tag, "w") as g:
        g.write(open(f).read())
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:8]]
for key in ("bench_under_trace", "pmc3"):
    try:
        out[key if key != "pmc3" else "bench_under_pmc"] = json.load(open(src + "/%s.json" % key))
    except Exception as e:  # noqa: BLE001
        out[key] = str(e)

pmc = {}
for f in sorted(glob.glob(src + "/pmc*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = klass(r["Kernel_Name"])
        if not k:
            continue
        d = pmc.setdefault(k, {}).setdefault(r["Counter_Name"], {"sum": 0.0, "launches": 0})
        d["sum"] += float(r["Counter_Value"])
        d["launches"] += 1
out["pmc"] = pmc

derived = {}
for k, t in pmc.items():
    g = lambda c: t.get(c, {}).get("sum", 0.0)  # noqa: E731
    fetch_b = g("FETCH_SIZE") * 1024.0 * 2.0
    write_b = g("WRITE_SIZE") * 1024.0
    derived[k] = {
        "launches": t.get("SQ_WAVES", t.get("FETCH_SIZE", {})).get("launches"),
        "waves": g("SQ_WAVES"),
        "valu_insts": g("SQ_INSTS_VALU"),
        "valu_lane_utilisation": (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0) if g("SQ_ACTIVE_INST_VALU") else None),
        "f64_add": g("SQ_INSTS_VALU_ADD_F64"), "f64_mul": g("SQ_INSTS_VALU_MUL_F64"), "f64_fma": g("SQ_INSTS_VALU_FMA_F64"),
        "f64_trans": g("SQ_INSTS_VALU_TRANS_F64"), "int32": g("SQ_INSTS_VALU_INT32"), "int64": g("SQ_INSTS_VALU_INT64"),
        "cvt": g("SQ_INSTS_VALU_CVT"), "salu": g("SQ_INSTS_SALU"), "smem": g("SQ_INSTS_SMEM"), "lds": g("SQ_INSTS_LDS"),
        "hbm_fetch_bytes_x2": fetch_b, "hbm_write_bytes": write_b,
    }
out["derived"] = derived

# HBM bytes per algorithmic byte of the dominant kernel, for bench.py's roofline.traffic
traffic = {}
b = out.get("bench_under_pmc")
if isinstance(b, dict) and "roofline" in b:
    name = b["roofline"]["kernel"]
    k = "trace_split" if name.endswith(("true>", ",1>")) else "trace_allinone"
    if k in derived and derived[k]["hbm_fetch_bytes_x2"]:
        alg = b["roofline"]["alg_bytes_per_launch"] * b["roofline"]["launches_per_step"] * b["steps"]
        hbm = derived[k]["hbm_fetch_bytes_x2"] + derived[k]["hbm_write_bytes"]
        m_cfg = re.search(r"BASELINE config (\d)", b["config"]["workload"])
        key = name if (m_cfg and m_cfg.group(1) == "4") else name + "@C" + (m_cfg.group(1) if m_cfg else "x")
        try:
            traffic = json.load(open("profiles/pmc_traffic.json"))  # one file for every config: config 4 under the bare kernel name
        except Exception:  # noqa: BLE001
            traffic = {}
        traffic[key] = {
            "hbm_bytes": hbm, "fetch_bytes_corrected_x2": derived[k]["hbm_fetch_bytes_x2"], "write_bytes": derived[k]["hbm_write_bytes"],
            "algorithmic_bytes": alg, "hbm_bytes_per_alg_byte": hbm / alg if alg else None,
            "workload": b["config"]["workload"],
            "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, all launches of this kernel "
                      "in the profiled command; FETCH_SIZE doubled per the gfx950 correction)" % tag}
        json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
out["pmc_traffic"] = traffic
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
print(json.dumps(derived, indent=1))
print(json.dumps(out.get("kernel_stats"), indent=1))
print(json.dumps(traffic, indent=1))
