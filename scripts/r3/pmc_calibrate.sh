#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3_pmc_calibrate}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/r3/probe_rows.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_0trace -- $CMD > $OUT/p0.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_1 -- $CMD > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_2 -- $CMD > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY > $OUT/calibration.txt
import csv, glob
for d, name in (("pmc_1", "FETCH_SIZE"), ("pmc_2", "WRITE_SIZE")):
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "probe" in r["Kernel_Name"]:
                print(name, r["Kernel_Name"][:60], r["Counter_Value"])
PY
grep "mode" $OUT/p0.log >> $OUT/calibration.txt
cat $OUT/calibration.txt
find $OUT -name "*.db" -delete
