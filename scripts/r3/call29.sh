#!/bin/bash
# round 3, call 29: the statistics of every hogwild test over 30 runs (tolerances = mean +- 5 sd)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python scripts/r3/stat_runs.py 30 > gpurun_out/r3ad_stat_runs.log 2>&1
echo "rc=$?"; cat gpurun_out/r3ad_stat_runs.log
