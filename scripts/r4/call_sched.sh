set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sgns_gpu.py tests/test_api_gpu.py tests/test_sgns_parity_gpu.py -x -q > gpurun_out/r4w_tests_sgns.log 2>&1 || { tail -30 gpurun_out/r4w_tests_sgns.log; exit 1; }
tail -3 gpurun_out/r4w_tests_sgns.log
timeout -k 10 300 python scripts/r4/time_sgns_sched.py 2>&1 | tee gpurun_out/r4w_time_sgns_sched.log
timeout -k 10 300 python scripts/fuzz_sgns.py 120 2>&1 | tail -2 | tee gpurun_out/r4w_fuzz_sgns.log
