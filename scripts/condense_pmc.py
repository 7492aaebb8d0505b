#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection CSVs under <dir>/pmc_*/ into <dir>/pmc_summary.json:
per kernel (name truncated) and counter: number of dispatches, mean and sum."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: [0, 0.0, []])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:90]
            k = (name, r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
            acc[k][2].append(float(r["Counter_Value"]))
    out[os.path.basename(d)] = {f"{k[0]} | {k[1]}": {"dispatches": v[0], "mean": v[1] / max(v[0], 1), "sum": v[1],
                                                              "median": sorted(v[2])[len(v[2]) // 2],
                                                              "values": v[2][:40]}
                                for k, v in sorted(acc.items()) if "n2v" in k[0]}
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
