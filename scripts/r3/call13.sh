#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_api_gpu.py tests/test_multirank_gpu.py tests/test_sgns_gpu.py -m gpu -q > gpurun_out/r3n_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3n_tests.log
tail -3 gpurun_out/r3n_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 700 python scripts/r3/e2e_cfg4.py batched 1.0 > gpurun_out/r3n_e2e_cfg4_batched.json 2> gpurun_out/r3n_e2e_cfg4_batched.err
echo "e2e rc=$?"; cat gpurun_out/r3n_e2e_cfg4_batched.json
