# round 6 (second session): the replays of long rows stepped in groups (walk_exact_wedge_replay_kernel) -- parity
# (also with every pairing on rows > 64 replayed), then timing against the lock-step kernel at both caps
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
N2V_REPLAY_GROUPS=1 timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py tests/test_margin_adversary_gpu.py -x -q > gpurun_out/r13a_tests_groups.log 2>&1 || { tail -40 gpurun_out/r13a_tests_groups.log; exit 1; }
tail -1 gpurun_out/r13a_tests_groups.log
N2V_REPLAY_GROUPS=1 N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge_forcereplay.so timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q -k "not 21000 and not 65535" > gpurun_out/r13a_tests_groups_forcereplay.log 2>&1 || { tail -40 gpurun_out/r13a_tests_groups_forcereplay.log; exit 1; }
tail -1 gpurun_out/r13a_tests_groups_forcereplay.log
N2V_REPLAY_GROUPS=1 timeout -k 10 200 python scripts/fuzz_walk.py 90 15 > gpurun_out/r13a_fuzz_groups.log 2>&1 || { tail -30 gpurun_out/r13a_fuzz_groups.log; exit 1; }
tail -1 gpurun_out/r13a_fuzz_groups.log
: > gpurun_out/r13a_time_groups.log
for gr in 0 1; do
  GROUPS=$gr CHECK=$gr PQ="0.5,2;4,0.25;3,0.7" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py groups$gr >> gpurun_out/r13a_time_groups.log 2>&1
done
for gr in 0 1; do
  GROUPS=$gr CHECK=$gr TRIM=10000 PQ="0.5,2;4,0.25;3,0.7;0.25,0.5" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py groups$gr >> gpurun_out/r13a_time_groups.log 2>&1
done
grep "G steps" gpurun_out/r13a_time_groups.log
