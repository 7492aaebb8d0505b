#!/bin/bash
# round 3, call 27: fused partition step for every (p, q) -- tests, steps/s, fuzz against the oracle
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_partitioned_gpu.py tests/test_multirank_gpu.py tests/test_capi_symbols.py -q -m "gpu or not gpu" > gpurun_out/r3aa_tests_partitioned.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r3aa_tests_partitioned.log
tail -3 gpurun_out/r3aa_tests_partitioned.log
timeout -k 10 300 python scripts/r3/time_partitioned.py > gpurun_out/r3aa_time_partitioned.log 2>&1
echo "time rc=$?"
cat gpurun_out/r3aa_time_partitioned.log
FUZZ_PARTITIONED=1 timeout -k 10 400 python scripts/fuzz_walk.py 200 11 > gpurun_out/r3aa_fuzz_partitioned.log 2>&1
echo "fuzz rc=$?"
tail -5 gpurun_out/r3aa_fuzz_partitioned.log
