#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_api_gpu.py tests/test_multirank_gpu.py -m gpu -q > gpurun_out/r3p_tests.log 2>&1
rc=$?; tail -2 gpurun_out/r3p_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_count.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r3p_time_count.log
timeout -k 10 300 python scripts/r3/e2e_cfg4.py batched 0.125 > gpurun_out/r3p_e2e_cfg4_eighth.json 2> gpurun_out/r3p_e2e.err; cat gpurun_out/r3p_e2e_cfg4_eighth.json
