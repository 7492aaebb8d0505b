# PMC passes of the exact biased kernels (one-launch through wedge_off, and from the per-edge records)
# on one graph: bytes past L2, L2 requests.  usage: bash scripts/r4/pmc_wedge.sh <tag> [GRAPH] [PQ]
set -x
TAG=${1:-r4f}; G=${2:-cfg4}; PQS=${3:-0.5,2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GRAPH=$G PQ=$PQS ROUNDS=""
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/scripts/r4/time_wedge2.py pmc > $OUT/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$OUT/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "walk_exact_wedge" not in k: continue
            short = "records" if "_rec_" in k else "wedge_off"
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (short, c), v in acc.items():
            top = max(v); keep = [x for x in v if x >= 0.5 * top]
            out[f"{short}|{c}"] = {"mean": sum(keep) / len(keep), "n": len(keep)}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +2M -delete
