# PMC pass over weighted exact walks (lanes + margins), weighted cfg 2, all walkers: instruction counts and waits
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7y}
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_pmc
OLD=0 BATCH=471785 PQ=${PQ:-"0.5,2.0"} timeout -k 10 900 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU} --output-format csv -d gpurun_out/${TAG}_pmc -o wm -- python3 scripts/r5/time_weighted_lanes.py > gpurun_out/${TAG}_pmc.log 2>&1 || { tail -30 gpurun_out/${TAG}_pmc.log; exit 1; }
f=$(find gpurun_out/${TAG}_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > gpurun_out/${TAG}_wm_pmc_summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if not ("weighted" in k): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[k] += 1
for k, d in agg.items():
    print(k, "dispatches", calls[k])
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v:.4g}  per dispatch {v / calls[k]:.4g}")
PY
cat gpurun_out/${TAG}_wm_pmc_summary.txt
rm -rf gpurun_out/${TAG}_pmc
