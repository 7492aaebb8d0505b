set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_partitioned_gpu.py tests/test_capi_symbols.py tests/test_multirank_gpu.py -x -q > gpurun_out/r5v_tests_part.log 2>&1 || { tail -40 gpurun_out/r5v_tests_part.log; exit 1; }
tail -3 gpurun_out/r5v_tests_part.log
FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 150 61 2>&1 | tail -3 | tee gpurun_out/r5v_fuzz_part.log
PQ="1.0,1.0;0.5,2.0;4.0,0.25;0.7,1.3" timeout -k 10 400 python scripts/r4/time_partitioned.py 2>&1 | grep "G steps" | tee gpurun_out/r5v_time_partitioned.log
