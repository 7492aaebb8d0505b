#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3c_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3c_tests.log
tail -8 gpurun_out/r3c_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 600 bash scripts/r3/pmc_batched.sh r3c_pmc_batched
