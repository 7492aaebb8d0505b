# GPU call 2 of round 2: gather ceiling (+ FETCH_SIZE calibration on known gather counts),
# the cfg 3/4/5 full-size tests, kernel-trace + PMC passes of bench.py on cfg 4 and cfg 3.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
hipcc --offload-arch=gfx950 -O3 -o scripts/micro/gather_ceiling scripts/micro/gather_ceiling.hip || exit 1
./scripts/micro/gather_ceiling > $OUT/gather_ceiling.log 2>&1 || exit 1
cat $OUT/gather_ceiling.log
timeout -k 10 900 python -m pytest tests/test_scale_cfg345_gpu.py -x -q --durations=5 > $OUT/tests_cfg345.log 2>&1
echo "tests_exit=$?" >> $OUT/tests_cfg345.log
tail -15 $OUT/tests_cfg345.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/gc_pmc_fetch -- $R/scripts/micro/gather_ceiling > $OUT/gc_pmc_fetch.log 2>&1) || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("$OUT/gc_pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = (r["Kernel_Name"].split("(")[0], )
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
with open("$OUT/gather_fetch_calibration.txt", "w") as o:
    for k, v in acc.items():
        o.write("%s dispatches=%d FETCH_SIZE_KB=%s\n" % (k[0], len(v), " ".join("%.0f" % x for x in v)))
print(open("$OUT/gather_fetch_calibration.txt").read())
PY
bash scripts/profile_r2.sh r02_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02_cfg3 --config cfg3 || exit 1
find $OUT -name "*.csv" -size +4M -delete
du -sh $R/gpurun_out
