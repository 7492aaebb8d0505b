set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_walk_gpu.py tests/test_partitioned_gpu.py tests/test_wedge_gpu.py -x -q > gpurun_out/r5p_tests_near.log 2>&1 || { tail -40 gpurun_out/r5p_tests_near.log; exit 1; }
tail -3 gpurun_out/r5p_tests_near.log
FUZZ_PARTITIONED=1 FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 180 31 2>&1 | tail -4 | tee gpurun_out/r5p_fuzz_near.log
FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 120 32 2>&1 | tail -4 | tee -a gpurun_out/r5p_fuzz_near.log
GRAPH=cfg4 PQ="0.7,3.0;3.0,0.7" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py near 2>&1 | grep "slots" | tee gpurun_out/r5p_time_near_cfg4.log
PQ="0.7,1.3;3.0,0.7" FORWARD=1 timeout -k 10 200 python scripts/r4/time_partitioned.py 2>&1 | grep "G steps" | tee gpurun_out/r5p_part_near.log
