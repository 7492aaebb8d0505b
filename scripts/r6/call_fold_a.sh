# round 6 (second session), call: folded lists and slots for the edges into wide rows -- tests, fuzz, timing
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q > gpurun_out/r12a_tests.log 2>&1 || { tail -60 gpurun_out/r12a_tests.log; exit 1; }
tail -3 gpurun_out/r12a_tests.log
timeout -k 10 300 python scripts/fuzz_walk.py 150 11 > gpurun_out/r12a_fuzz.log 2>&1 || { tail -30 gpurun_out/r12a_fuzz.log; exit 1; }
tail -2 gpurun_out/r12a_fuzz.log
PQ="0.5,2;4,0.25;3,0.7" REPS=3 timeout -k 10 400 python scripts/r6/time_variant.py r12a_fold > gpurun_out/r12a_time_fold_cap100000.log 2>&1 || { tail -30 gpurun_out/r12a_time_fold_cap100000.log; exit 1; }
grep "G steps" gpurun_out/r12a_time_fold_cap100000.log
TRIM=10000 PQ="0.5,2;4,0.25;3,0.7" REPS=3 timeout -k 10 400 python scripts/r6/time_variant.py r12a_fold > gpurun_out/r12a_time_fold_cap10000.log 2>&1 || { tail -30 gpurun_out/r12a_time_fold_cap10000.log; exit 1; }
grep "G steps" gpurun_out/r12a_time_fold_cap10000.log
