#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/r04_profile_n3.sh) into profiles/<tag>_summary.json: for both forms of the primary
segment per kernel the device time and the PMC sums over the two frames of the probe; and the FETCH_SIZE control (what the counter
reports for a pointer chase whose record count is known)."""
import csv, glob, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_n3"
src = "gpurun_out/prof_" + tag
out = {"tag": tag}
for mode in ("coop", "lane"):
    o = {"workload": open("%s/%s.probe.txt" % (src, mode)).read().strip().splitlines()[-1]}
    for f in glob.glob("%s/%s/trace/**/*_kernel_stats.csv" % (src, mode), recursive=True):
        rows = list(csv.DictReader(open(f)))
        open("profiles/%s_%s_kernel_stats.csv" % (tag, mode), "w").write(open(f).read())
        o["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:6]]
    pmc = {}
    for f in sorted(glob.glob("%s/%s/pmc*/**/*_counter_collection.csv" % (src, mode), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "ptk::" not in name:
                continue
            pmc.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
            pmc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    o["pmc_sum_over_the_probe"] = pmc
    times = {r["Name"].split("(")[0].replace("void ", ""): float(r["TotalDurationNs"]) * 1e-9 for r in o.get("kernel_stats", [])}
    der = {}
    for k, t in pmc.items():
        g = lambda c: t.get(c, 0.0)  # noqa: E731
        d = {"seconds": times.get(k), "hbm_bytes_fetch_x2_plus_write": g("FETCH_SIZE") * 2048.0 + g("WRITE_SIZE") * 1024.0,
             "hbm_bytes_fetch_x1_plus_write": g("FETCH_SIZE") * 1024.0 + g("WRITE_SIZE") * 1024.0}
        if times.get(k):
            d["hbm_gb_per_s_x2"] = d["hbm_bytes_fetch_x2_plus_write"] / times[k] / 1e9
            d["hbm_gb_per_s_x1"] = d["hbm_bytes_fetch_x1_plus_write"] / times[k] / 1e9
        if g("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
            d["valu_insts"] = g("SQ_INSTS_VALU")
        if g("TCC_REQ_sum"):
            d["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
        if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
            d["vl1d_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
        if g("SQ_WAVE_CYCLES"):
            d["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") else None
        if g("GRBM_GUI_ACTIVE") and g("TA_TA_BUSY_sum"):
            d["ta_busy_cycles_per_gui_cycle_summed_over_tas"] = g("TA_TA_BUSY_sum") / g("GRBM_GUI_ACTIVE")
        der[k] = d
    o["derived"] = der
    out[mode] = o
# control
ctl = {}
try:
    bench = json.load(open(src + "/control.FETCH_SIZE.json"))["results"]
    for what in ("FETCH_SIZE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum"):
        rows = []
        for f in sorted(glob.glob("%s/control/%s/**/*_counter_collection.csv" % (src, what), recursive=True)):
            rows += list(csv.DictReader(open(f)))
        disp = {}
        for r in rows:
            disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(disp)
        # the bench launches every configuration twice (warm-up + timed), in the order of its result list
        for k, b in enumerate(bench):
            pair = ids[2 * k: 2 * k + 2]
            if len(pair) < 2:
                break
            c = disp[pair[1]]
            key = "%d MiB, %d waves/SIMD, %d loads, %d lanes" % (b["table_mib"], b["waves_per_simd"], b["loads_per_visit"], b["lanes"])
            e = ctl.setdefault(key, {"record_visits": 256 * b["waves_per_simd"] * 4 * b["lanes"] * 2000})
            e.update(c)
    for key, e in ctl.items():
        v = e["record_visits"]
        if "FETCH_SIZE" in e:
            e["fetch_size_bytes_per_visit_as_reported"] = e["FETCH_SIZE"] * 1024.0 / v
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in e:
            e["tcp_cache_accesses_per_visit"] = e["TCP_TOTAL_CACHE_ACCESSES_sum"] / v
            e["tcp_tcc_read_req_per_visit"] = e.get("TCP_TCC_READ_REQ_sum", 0.0) / v
except Exception as ex:  # noqa: BLE001
    ctl = {"error": repr(ex)}
out["fetch_size_control"] = ctl
json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
for mode in ("coop", "lane"):
    print(mode, json.dumps(out[mode]["derived"], indent=1))
    print(out[mode]["workload"])
print(json.dumps({k: v for k, v in ctl.items() if "512 MiB" in k or "1 MiB" in k}, indent=1) if isinstance(ctl, dict) else ctl)
