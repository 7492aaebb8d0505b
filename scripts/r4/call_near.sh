set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_walk_gpu.py tests/test_partitioned_gpu.py tests/test_wedge_gpu.py tests/test_scale_cfg345_gpu.py -x -q > gpurun_out/r5s_tests_near.log 2>&1 || { tail -40 gpurun_out/r5s_tests_near.log; exit 1; }
tail -3 gpurun_out/r5s_tests_near.log
FUZZ_PARTITIONED=1 FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 150 41 2>&1 | tail -4 | tee gpurun_out/r5s_fuzz_near.log
timeout -k 10 300 python scripts/fuzz_walk.py 120 42 2>&1 | tail -4 | tee -a gpurun_out/r5s_fuzz_near.log
GRAPH=cfg4 PQ="0.7,3.0;1.3,1.3;3.0,0.7;0.3,0.7;3.0,1.0" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py near 2>&1 | grep "slots" | tee gpurun_out/r5s_time_near_cfg4.log
GRAPH=cfg2 PQ="0.7,3.0;3.0,0.7;0.5,2.0" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py near 2>&1 | grep "slots" | tee -a gpurun_out/r5s_time_near_cfg4.log
