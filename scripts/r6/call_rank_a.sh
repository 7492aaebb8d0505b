# round 6 (second session): the slot of a rank by ONE search of the list (unlisted_from_top) instead of the fixed-point
# iteration -- parity (also with every pairing on rows > 64 replayed), fuzz, timing at both caps
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py tests/test_margin_adversary_gpu.py -x -q > gpurun_out/r13f_tests.log 2>&1 || { tail -40 gpurun_out/r13f_tests.log; exit 1; }
tail -1 gpurun_out/r13f_tests.log
for v in forcereplay forcereplay2; do
N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge_$v.so timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q -k "not 21000 and not 65535" > gpurun_out/r13f_tests_$v.log 2>&1 || { tail -40 gpurun_out/r13f_tests_$v.log; exit 1; }
tail -1 gpurun_out/r13f_tests_$v.log
done
FUZZ_PQ=two timeout -k 10 200 python scripts/fuzz_walk.py 60 16 > gpurun_out/r13f_fuzz_two.log 2>&1 || { tail -30 gpurun_out/r13f_fuzz_two.log; exit 1; }
tail -1 gpurun_out/r13f_fuzz_two.log
timeout -k 10 200 python scripts/fuzz_walk.py 90 17 > gpurun_out/r13f_fuzz.log 2>&1 || { tail -30 gpurun_out/r13f_fuzz.log; exit 1; }
tail -1 gpurun_out/r13f_fuzz.log
: > gpurun_out/r13f_time.log
for v in base abl4; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib PQ="4,0.25;3,0.7;0.5,2;0.25,0.5" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r13f_time.log 2>&1
done
TRIM=10000 PQ="4,0.25;3,0.7;0.5,2;0.25,0.5;4,2" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py base >> gpurun_out/r13f_time.log 2>&1
grep "G steps" gpurun_out/r13f_time.log
