# counters of the slots kernel at (0.5, 2) on cfg 3 trimmed at 10 000 against 100 000: what do the launches at the
# reference's cap wait for?   usage: bash scripts/r5/pmc_cap.sh <tag>
set -x
TAG=${1:-r7w}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GRAPH=cfg3 PQ="0.5,2.0;4.0,0.25" ROUNDS=""
for T in 10000 100000; do
  export TRIM=$T
  i=0
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d $OUT/t${T}_p$i -- python3 $R/scripts/r4/time_wedge2.py pmc > $OUT/t${T}_p$i.log 2>&1 || echo "pass $T $i failed"
  done
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$OUT/t*_p*/")):
    trim = d.split("/")[-2].split("_")[0]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        order = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "walk_exact_wedge_slots_kernel" not in k: continue
            inst = k.split("walk_exact_wedge_slots_kernel")[1][:3]
            acc[(inst, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (inst, c), v in acc.items():
            top = max(v); keep = [x for x in v if x >= 0.5 * top]
            out[f"{trim}|slots{inst}|{c}"] = sum(keep) / len(keep)
steps = 8.388608e8
for k in sorted(out):
    extra = ""
    if k.endswith("FETCH_SIZE"): extra = f"  = {out[k] * 1024 / 64 / steps:.3f} sectors per step"
    if k.endswith("TCC_REQ_sum"): extra = f"  = {out[k] / steps:.3f} L2 requests per step"
    print(k, f"{out[k]:.4g}", extra)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +2M -delete
