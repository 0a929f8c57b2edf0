#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel.  usage: pmc_summary.py <csv>... -> JSON on stdout:
{kernel: {counter: sum over dispatches, "dispatches": n}}"""
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(s in k for s in ("nxz", "nxzl")):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
print(json.dumps({k: dict(v, dispatches=len(disp[k])) for k, v in agg.items()}, indent=1, sort_keys=True))
