#!/bin/bash
# round 3, call 33: the profile of the last build (fast mode is a new sampler): kernel trace + PMC passes
set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 60; echo "... profiling $(date +%T)"; done ) &
HB=$!
timeout -k 10 1100 bash scripts/r3/profile_r3.sh r3ag_cfg4 > gpurun_out/r3ag_profile.log 2>&1
rc=$?
kill $HB
echo "profile rc=$rc"; tail -3 gpurun_out/r3ag_profile.log
