"""counters of the walk_uniform_kernel dispatches of a `rocprofv3 --pmc` run of placement_pmc.py, averaged
per placement (1 warm-up + REPS timed launches each; the warm-up dropped):  condense.py <dir> [REPS]"""
import csv, glob, os, sys, collections
d, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = collections.defaultdict(dict)  # dispatch id -> counter -> value
dur = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "walk_uniform_kernel" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
ids = sorted(rows)
per = 1 + reps
for p in range(len(ids) // per):
    grp = ids[p * per + 1:(p + 1) * per]
    names = sorted(rows[grp[0]])
    avg = {n: sum(rows[i][n] for i in grp) / len(grp) for n in names}
    ms = sum(dur.get(i, 0.0) for i in grp) / len(grp)
    print(f"placement {p}: kernel {ms:.2f} ms  " + "  ".join(f"{n}={avg[n]:.4g}" for n in names))
