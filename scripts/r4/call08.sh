set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
timeout -k 10 300 python -m pytest tests/test_walk_gpu.py tests/test_wedge_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/r4l_tests.log 2>&1 || { tail -30 gpurun_out/r4l_tests.log; exit 1; }
tail -2 gpurun_out/r4l_tests.log
timeout -k 10 200 python scripts/fuzz_walk.py 90 4006 > gpurun_out/r4l_fuzz.log 2>&1 || { tail -20 gpurun_out/r4l_fuzz.log; exit 1; }
tail -1 gpurun_out/r4l_fuzz.log
GRAPH=cfg4 PQ="0.7,3;1.3,1.3;3,1;3,0.7;0.5,2" ROUNDS="" timeout -k 10 600 python scripts/r4/time_wedge2.py nd 2>&1 | grep -v amdgpu | tee gpurun_out/r4l_time_nondyadic_cfg4.log
HUB_ROWS=auto timeout -k 10 300 python scripts/r3/hogwild_auc_runs.py 5 default 0 2>&1 | grep -v amdgpu | tee gpurun_out/r4l_hogwild_auc_auto.log
