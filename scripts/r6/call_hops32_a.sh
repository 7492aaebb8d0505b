# round 6 (second session): 32-byte hop entries (hop entry + first half of the edge's slot) -- parity, timing
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
N2V_HOPS32=1 timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r14a_tests_hops32.log 2>&1 || { tail -40 gpurun_out/r14a_tests_hops32.log; exit 1; }
tail -1 gpurun_out/r14a_tests_hops32.log
N2V_HOPS32=1 timeout -k 10 200 python scripts/fuzz_walk.py 90 22 > gpurun_out/r14a_fuzz_hops32.log 2>&1 || { tail -30 gpurun_out/r14a_fuzz_hops32.log; exit 1; }
tail -1 gpurun_out/r14a_fuzz_hops32.log
: > gpurun_out/r14a_time.log
for rep in 1 2; do
for h in 0 1; do
  N2V_HOPS32=$h TRIM=10000 PQ="0.5,2;4,0.25;3,0.7;0.25,0.5" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py hops32_$h >> gpurun_out/r14a_time.log 2>&1
done
done
for h in 0 1; do
  N2V_HOPS32=$h PQ="0.5,2;4,0.25;3,0.7" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py hops32_$h >> gpurun_out/r14a_time.log 2>&1
done
grep "G steps" gpurun_out/r14a_time.log
