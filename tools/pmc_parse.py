#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel.  usage: pmc_parse.py <dir-with-*_counter_collection.csv> [filter]"""
import collections, csv, glob, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "splat"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if flt not in k: continue
    print(k, "launches", len(dur[k]), "avg_us", round(sum(dur[k]) / max(1, len(dur[k])) / 1e3, 1))
    for c, vals in sorted(v.items()):
        print(f"    {c:32s} {sum(vals)/len(vals):18.1f}")
