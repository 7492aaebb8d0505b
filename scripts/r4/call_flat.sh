set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_walk_gpu.py tests/test_wedge_gpu.py tests/test_scale_props_gpu.py tests/test_partitioned_gpu.py -x -q > gpurun_out/r6c_tests_flat.log 2>&1 || { tail -40 gpurun_out/r6c_tests_flat.log; exit 1; }
tail -2 gpurun_out/r6c_tests_flat.log
FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 100 91 2>&1 | tail -3 | tee gpurun_out/r6c_fuzz_flat.log
GRAPH=cfg4 PQ="0.5,2.0;0.25,0.5" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py flat 2>&1 | grep "slots" | tee gpurun_out/r6c_time_flat_cfg4.log
