# round 6 (second session): the stored row sums of (p, q) that are not dyadic -- tests, fuzz, timing at both caps
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py tests/test_margin_adversary_gpu.py -x -q > gpurun_out/r12e_tests.log 2>&1 || { tail -60 gpurun_out/r12e_tests.log; exit 1; }
tail -3 gpurun_out/r12e_tests.log
FUZZ_PQ=rational timeout -k 10 300 python scripts/fuzz_walk.py 100 12 > gpurun_out/r12e_fuzz_rational.log 2>&1 || { tail -30 gpurun_out/r12e_fuzz_rational.log; exit 1; }
tail -1 gpurun_out/r12e_fuzz_rational.log
PQ="3,0.7;0.7,3;1.3,1.3" REPS=3 timeout -k 10 400 python scripts/r6/time_variant.py r12e > gpurun_out/r12e_time_rowsums_cap100000.log 2>&1 || { tail -30 gpurun_out/r12e_time_rowsums_cap100000.log; exit 1; }
grep "G steps" gpurun_out/r12e_time_rowsums_cap100000.log
TRIM=10000 PQ="3,0.7;0.7,3" REPS=3 timeout -k 10 400 python scripts/r6/time_variant.py r12e > gpurun_out/r12e_time_rowsums_cap10000.log 2>&1 || { tail -30 gpurun_out/r12e_time_rowsums_cap10000.log; exit 1; }
grep "G steps" gpurun_out/r12e_time_rowsums_cap10000.log
