#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_gpu.py -m gpu -q -s > gpurun_out/r3k_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3k_tests.log
tail -4 gpurun_out/r3k_tests.log; grep "norm of syn0" gpurun_out/r3k_tests.log
[ $rc -le 1 ] || exit 1
for h in 0 4096 65536; do
  HUB_ROWS=$h timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "batched=False" 
done | tee gpurun_out/r3k_time_hub_rows_cfg3.log
for h in 4096 65536; do
  HUB_ROWS=$h timeout -k 10 300 python scripts/r3/hogwild_auc_runs.py 5 default 0 2>&1 | grep -v "^/opt" 
done | tee gpurun_out/r3k_hogwild_auc_hub_rows.log | grep "hub_rows\|mean"
