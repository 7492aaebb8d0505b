# first run of the two-pass biased kernel: parity on the small suites + fuzz, then timing on cfg 4
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_wedge_gpu.py tests/test_edge_cases_gpu.py tests/test_api_gpu.py -x -q -m gpu > gpurun_out/r4b_tests.log 2>&1 || { tail -30 gpurun_out/r4b_tests.log; exit 1; }
tail -3 gpurun_out/r4b_tests.log
timeout -k 10 200 python scripts/fuzz_walk.py 90 4001 > gpurun_out/r4b_fuzz.log 2>&1 || { tail -20 gpurun_out/r4b_fuzz.log; exit 1; }
tail -2 gpurun_out/r4b_fuzz.log
GRAPH=cfg4 PQ="0.5,2;4,0.25;4,2" ROUNDS="4;2;6" timeout -k 10 500 python scripts/r4/time_wedge2.py w7 > gpurun_out/r4b_time_cfg4.log 2>&1 || { tail -20 gpurun_out/r4b_time_cfg4.log; exit 1; }
cat gpurun_out/r4b_time_cfg4.log
