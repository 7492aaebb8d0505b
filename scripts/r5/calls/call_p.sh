set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_partitioned_gpu.py tests/test_alias_trim_fast_gpu.py -x -q > gpurun_out/r7r_tests.log 2>&1 || { tail -40 gpurun_out/r7r_tests.log; exit 1; }
tail -2 gpurun_out/r7r_tests.log
timeout -k 10 300 python scripts/fuzz_walk.py 120 91 2>&1 | tail -2 | tee gpurun_out/r7r_fuzz.log
OLD=1 BIG=47104 BATCH=47104 KINDS=fp32,fp64 PQ="0.5,2.0;3.0,0.7" timeout -k 10 400 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "wave " | tee gpurun_out/r7r_time_wave.log
timeout -k 10 200 python scripts/time_weighted.py 2>&1 | grep Msteps | tee -a gpurun_out/r7r_time_wave.log
