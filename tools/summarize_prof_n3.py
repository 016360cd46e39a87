#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_n3.sh) into profiles/<tag>_summary.json: per kernel the device time of the
trace pass and the PMC sums over both frames of the probe (HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE, KiB units)."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_n3"
src = "gpurun_out/prof_" + tag
out = {"tag": tag, "workload": open(src + "/probe.txt").read().strip()}
for f in glob.glob(src + "/trace/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    open("profiles/%s_kernel_stats.csv" % tag, "w").write(open(f).read())
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:8]]
pmc = {}
for f in sorted(glob.glob(src + "/pmc*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ptk::" not in name:
            continue
        pmc.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
        pmc[name][r["Counter_Name"]] += float(r["Counter_Value"])
out["pmc_sum_over_the_probe"] = pmc
der = {}
times = {r["Name"].split("(")[0].replace("void ", ""): float(r["TotalDurationNs"]) * 1e-9 for r in out.get("kernel_stats", [])}
for k, t in pmc.items():
    g = lambda c: t.get(c, 0.0)  # noqa: E731
    hbm = g("FETCH_SIZE") * 2048.0 + g("WRITE_SIZE") * 1024.0
    d = {"hbm_bytes": hbm, "seconds": times.get(k)}
    if times.get(k):
        d["hbm_gb_per_s"] = hbm / times[k] / 1e9
        d["frac_of_8TBps"] = hbm / times[k] / 8e12
    if g("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
        d["valu_insts"] = g("SQ_INSTS_VALU")
    if g("TCC_REQ_sum"):
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
    der[k] = d
out["derived"] = der
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
print(json.dumps(der, indent=1))
print(out["workload"])
