#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_gpu.py -m gpu -q > gpurun_out/r3m_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r3m_tests.log
tail -3 gpurun_out/r3m_tests.log
[ $rc -le 1 ] || exit 1
timeout -k 10 300 python scripts/r3/time_batched.py cfg3 128 2>&1 | grep "batched=" | tee gpurun_out/r3m_time_sgns_cfg3.log
timeout -k 10 300 python scripts/r3/time_partitioned.py 2>&1 | grep "p=" | tee gpurun_out/r3m_time_partitioned.log
