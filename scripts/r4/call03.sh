# wedge slots: parity (tests + fuzz) then timing on cfg 4 / cfg 3
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_wedge_gpu.py tests/test_edge_cases_gpu.py tests/test_api_gpu.py tests/test_delta_sync_gpu.py -x -q -m gpu > gpurun_out/r4e_tests.log 2>&1 || { tail -30 gpurun_out/r4e_tests.log; exit 1; }
tail -3 gpurun_out/r4e_tests.log
timeout -k 10 200 python scripts/fuzz_walk.py 90 4002 > gpurun_out/r4e_fuzz.log 2>&1 || { tail -20 gpurun_out/r4e_fuzz.log; exit 1; }
tail -2 gpurun_out/r4e_fuzz.log
GRAPH=cfg4 PQ="0.5,2;4,0.25;4,2;2,1" ROUNDS="" timeout -k 10 500 python scripts/r4/time_wedge2.py slots > gpurun_out/r4e_time_cfg4.log 2>&1 || { tail -20 gpurun_out/r4e_time_cfg4.log; exit 1; }
cat gpurun_out/r4e_time_cfg4.log
GRAPH=cfg3 PQ="0.5,2;4,0.25" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py slots > gpurun_out/r4e_time_cfg3.log 2>&1 || { tail -20 gpurun_out/r4e_time_cfg3.log; exit 1; }
cat gpurun_out/r4e_time_cfg3.log
