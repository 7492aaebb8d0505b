set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_partitioned_gpu.py tests/test_capi_symbols.py tests/test_multirank_gpu.py -x -q > gpurun_out/r5a_tests_part.log 2>&1 || { tail -40 gpurun_out/r5a_tests_part.log; exit 1; }
tail -3 gpurun_out/r5a_tests_part.log
timeout -k 10 400 python scripts/r4/time_partitioned.py 2>&1 | tee gpurun_out/r5a_time_partitioned.log
