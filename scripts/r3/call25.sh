#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r4b_tests.log 2>&1; tail -2 gpurun_out/r4b_tests.log
timeout -k 10 900 bash scripts/r3/pmc_batched.sh r3z_pmc_batched > gpurun_out/r4b_pmc.log 2>&1
grep "batched=" gpurun_out/r3z_pmc_batched/p0.log
