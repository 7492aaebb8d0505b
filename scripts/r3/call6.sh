#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r3f_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3f_tests.log
tail -5 gpurun_out/r3f_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 900 bash scripts/r3/pmc_batched.sh r3f_pmc_batched
grep "batched=" gpurun_out/r3f_pmc_batched/p0.log
