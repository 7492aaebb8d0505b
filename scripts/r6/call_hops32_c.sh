# round 6 (second session): 32-byte hop entries as instances of their own (template parameter) -- parity, timing
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for h in 1 0; do
N2V_HOPS32=$h timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r14j_tests_$h.log 2>&1 || { tail -40 gpurun_out/r14j_tests_$h.log; exit 1; }
tail -1 gpurun_out/r14j_tests_$h.log
done
: > gpurun_out/r14j_time.log
PQ="0.5,2;4,0.25;3,0.7" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py cap100000 2>&1 | grep "G steps" >> gpurun_out/r14j_time.log
for h in 0 1; do
  N2V_HOPS32=$h TRIM=10000 PQ="0.5,2;4,0.25;3,0.7;0.25,0.5" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py hops32_$h 2>&1 | grep "G steps" >> gpurun_out/r14j_time.log
done
cat gpurun_out/r14j_time.log
