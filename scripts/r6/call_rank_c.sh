# round 6 (second session): the walk down to the next free slot bounded at four steps (then one search) -- parity, timing
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py tests/test_margin_adversary_gpu.py -x -q > gpurun_out/r13r_tests.log 2>&1 || { tail -40 gpurun_out/r13r_tests.log; exit 1; }
tail -1 gpurun_out/r13r_tests.log
FUZZ_PQ=two timeout -k 10 200 python scripts/fuzz_walk.py 60 18 > gpurun_out/r13r_fuzz_two.log 2>&1 || { tail -30 gpurun_out/r13r_fuzz_two.log; exit 1; }
tail -1 gpurun_out/r13r_fuzz_two.log
timeout -k 10 200 python scripts/fuzz_walk.py 90 19 > gpurun_out/r13r_fuzz.log 2>&1 || { tail -30 gpurun_out/r13r_fuzz.log; exit 1; }
tail -1 gpurun_out/r13r_fuzz.log
FUZZ_PQ=rational timeout -k 10 200 python scripts/fuzz_walk.py 60 20 > gpurun_out/r13r_fuzz_rational.log 2>&1 || { tail -30 gpurun_out/r13r_fuzz_rational.log; exit 1; }
tail -1 gpurun_out/r13r_fuzz_rational.log
: > gpurun_out/r13r_time.log
PQ="4,0.25;3,0.7;0.25,0.5;0.5,2" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py new >> gpurun_out/r13r_time.log 2>&1
TRIM=10000 PQ="4,0.25;3,0.7;0.25,0.5;0.5,2" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py new >> gpurun_out/r13r_time.log 2>&1
grep "G steps" gpurun_out/r13r_time.log
