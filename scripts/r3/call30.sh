#!/bin/bash
# round 3, call 30: are there rare outlier runs of the hogwild trainer? (one suite run had
# Procrustes 0.806 between two hogwild runs whose 30-run distribution is 0.958 +- 0.001)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python scripts/r3/stat_outliers.py 400 > gpurun_out/r3ad_stat_outliers.log 2>&1
echo "rc=$?"; tail -30 gpurun_out/r3ad_stat_outliers.log
