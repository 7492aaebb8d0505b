#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sgns_batched_gpu.py -m gpu -q > gpurun_out/r3r_tests.log 2>&1; tail -2 gpurun_out/r3r_tests.log
timeout -k 10 400 python scripts/fuzz_sgns.py 300 31 > gpurun_out/r3r_fuzz_sgns.log 2>&1; tail -2 gpurun_out/r3r_fuzz_sgns.log
