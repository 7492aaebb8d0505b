set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ranked_gpu.py tests/test_walk_gpu.py tests/test_capi_symbols.py -x -q > gpurun_out/r4x_tests_ranked.log 2>&1 || { tail -40 gpurun_out/r4x_tests_ranked.log; exit 1; }
tail -3 gpurun_out/r4x_tests_ranked.log
GRAPH=cfg2 timeout -k 10 300 python scripts/r4/time_ranked.py 2>&1 | tee gpurun_out/r4x_time_ranked_cfg2.log
GRAPH=cfg4 timeout -k 10 600 python scripts/r4/time_ranked.py 2>&1 | tee gpurun_out/r4x_time_ranked_cfg4.log
