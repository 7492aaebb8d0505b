# round 5, call q: 8-ary search of long wedge lists -- tests, fuzz, cfg 3 at cap 100 000, cfg 4 at 10 000
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_walk_gpu.py tests/test_partitioned_gpu.py tests/test_fast_unit_gpu.py tests/test_scale_props_gpu.py -x -q > gpurun_out/r7s_tests.log 2>&1 || { tail -40 gpurun_out/r7s_tests.log; exit 1; }
tail -2 gpurun_out/r7s_tests.log
FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 100 95 2>&1 | tail -2 | tee gpurun_out/r7s_fuzz.log
FUZZ_PQ=two timeout -k 10 200 python scripts/fuzz_walk.py 60 96 2>&1 | tail -2 | tee -a gpurun_out/r7s_fuzz.log
GRAPH=cfg3 TRIM=100000 PQ="0.5,2.0;4.0,0.25;3.0,0.7;4.0,2.0" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py kary 2>&1 | grep "+ slots\|mode" | tee gpurun_out/r7s_time_kary.log
GRAPH=cfg4 TRIM=10000 PQ="0.5,2.0;4.0,0.25;3.0,0.7;4.0,2.0" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py kary 2>&1 | grep "+ slots" | tee -a gpurun_out/r7s_time_kary.log
