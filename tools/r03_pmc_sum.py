#!/usr/bin/env python3
"""Sums a rocprofv3 counter_collection.csv per kernel: python tools/r03_pmc_sum.py <csv>"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0))):
    if "ptk::" not in k:
        continue
    print("%-60s calls %3d  %s" % (k[-60:], len(calls[k]), "  ".join("%s %.4g" % (a, b) for a, b in sorted(v.items()))))
