# round 5, call b: the mixed wedge table -- tests, fuzz, cfg 3 / cfg 4 at the reference's default trim cap
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_walk_gpu.py tests/test_partitioned_gpu.py tests/test_fast_unit_gpu.py -x -q > gpurun_out/r7b_tests_mixed.log 2>&1 || { tail -40 gpurun_out/r7b_tests_mixed.log; exit 1; }
tail -3 gpurun_out/r7b_tests_mixed.log
FUZZ_PARTITIONED=1 timeout -k 10 300 python scripts/fuzz_walk.py 120 71 2>&1 | tail -4 | tee gpurun_out/r7b_fuzz_mixed.log
FUZZ_PQ=two timeout -k 10 200 python scripts/fuzz_walk.py 60 72 2>&1 | tail -4 | tee -a gpurun_out/r7b_fuzz_mixed.log
for T in 10000 100000; do
GRAPH=cfg3 TRIM=$T PQ="0.5,2.0;3.0,0.7;4.0,0.25" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py trim$T 2>&1 | grep "trim" | tee -a gpurun_out/r7b_time_mixed.log
done
for T in 10000 100000; do
GRAPH=cfg4 TRIM=$T PQ="0.5,2.0;3.0,0.7;4.0,0.25" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py trim$T 2>&1 | grep "trim" | tee -a gpurun_out/r7b_time_mixed.log
done
