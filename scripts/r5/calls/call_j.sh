# round 5, call j: kernel trace of the weighted lane walk (which launches take the time?)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r7j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export N2V_WLANES_SHORT=100000000 OLD=0 BATCH=47104 BIG=47104 KINDS=fp32 PQ="0.5,2.0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/r5/time_weighted_lanes.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
grep "steps/s" $OUT/trace.log
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -8 $f | cut -c1-260; done
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "walk_weighted_step" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("walk_weighted_step launches", len(d), "ms: first 10", [round(x, 2) for x in d[:10]], "steps 70..80 of the last walk", [round(x, 2) for x in d[-10:]])
print("sum ms", sum(d), "max", max(d))
PY
find $OUT -name "*.csv" -size +3M -delete
