#!/bin/bash
# round 3, call 26: where a partitioned step spends its time (kernel trace of time_partitioned.py)
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_part -o part -- python3 $GRAFT_REPO_ROOT/scripts/r3/time_partitioned.py > $GRAFT_REPO_ROOT/gpurun_out/r3ab_prof_part.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_part/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = open("gpurun_out/r3ab_part_kernel_stats.txt", "w")
for r in rows[:25]:
    line = f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:90]}'
    print(line); out.write(line + "\n")
print("total kernel ms", tot / 1e6); out.write(f"total kernel ms {tot/1e6}\n")
# the step kernel by part (calls rotate over the 8 parts)
f = glob.glob("gpurun_out/prof_part/**/*kernel_trace.csv", recursive=True)
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f[0]))
     if r["Kernel_Name"].startswith("void n2v::partition_step_unit")]
d.sort()
d = [x[1] for x in d]
for case in range(len(d) // 160):
    c = d[case * 160:(case + 1) * 160]
    line = f"case {case}: per part mean us " + " ".join(f"{sum(c[p::8]) / len(c[p::8]) / 1e3:8.1f}" for p in range(8)) + f"  max {max(c)/1e3:.1f}"
    print(line); out.write(line + "\n")
PY
tail -4 gpurun_out/r3ab_prof_part.log
rm -rf gpurun_out/prof_part
