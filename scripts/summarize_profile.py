#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ rocprofv3 run into profiles/<tag>_*.{csv,json}."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "walk_exact"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag, "kernel_filter": kernel}
stats = glob.glob(f"{src}/trace/*/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:12]:
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    for r in rows:
        if kernel in r["Name"]:
            out["kernel_trace"] = {"name": r["Name"][:120], "calls": int(r["Calls"]),
                                   "avg_ms": float(r["AverageNs"]) / 1e6,
                                   "pct": float(r["Percentage"])}
            break
for kind, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(f"{src}/{kind}/*/*counter_collection.csv")
    if not files:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == cname]
    if vals:
        out[cname + "_KB_per_launch_raw"] = sum(vals) / len(vals)
if "FETCH_SIZE_KB_per_launch_raw" in out:
    f, w = out["FETCH_SIZE_KB_per_launch_raw"], out.get("WRITE_SIZE_KB_per_launch_raw", 0.0)
    # MI355X_MICROARCH.md "HBM": counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the
    # bytes of wide coalesced reads -> doubled (upper bound for narrower accesses).
    out["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    out["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024.0
    out["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"
bench = os.path.join(src, "bench.json")
if os.path.exists(bench):
    try:
        out["bench"] = json.loads(open(bench).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        pass
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench"}, indent=1))
