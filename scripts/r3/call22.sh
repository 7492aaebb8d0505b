#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 bash scripts/r3/profile_r3.sh r3z_cfg4 > gpurun_out/r3z_profile.log 2>&1
tail -3 gpurun_out/r3z_profile.log
