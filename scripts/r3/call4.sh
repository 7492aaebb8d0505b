#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3d_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3d_tests.log
tail -8 gpurun_out/r3d_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 > gpurun_out/r3d_time_batched_cfg3.log 2>&1 && cat gpurun_out/r3d_time_batched_cfg3.log
timeout -k 10 300 python scripts/r3/hogwild_auc_runs.py 5 batched 0 > gpurun_out/r3d_hogwild_auc_batched.log 2>&1 && tail -7 gpurun_out/r3d_hogwild_auc_batched.log
