# round 6 (second session): the replay of the "other overfull" arrangement without list loads -- parity with every
# pairing on rows of more than 64 slots REPLAYED (-DN2V_FORCE_REPLAY builds), then timing; the big instances at six waves
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in forcereplay forcereplay2; do
N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge_$v.so timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_long_lists_gpu.py tests/test_walk_gpu.py -x -q -k "not 21000 and not 65535" > gpurun_out/r12i_tests_$v.log 2>&1 || { tail -40 gpurun_out/r12i_tests_$v.log; exit 1; }
tail -1 gpurun_out/r12i_tests_$v.log
done
N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge_forcereplay.so FUZZ_PQ=two timeout -k 10 200 python scripts/fuzz_walk.py 60 13 > gpurun_out/r12i_fuzz_forcereplay.log 2>&1 || { tail -30 gpurun_out/r12i_fuzz_forcereplay.log; exit 1; }
tail -1 gpurun_out/r12i_fuzz_forcereplay.log
N2V_HIP_LIB=$PWD/build_variants/libn2v_wedge_forcereplay2.so timeout -k 10 200 python scripts/fuzz_walk.py 60 14 > gpurun_out/r12i_fuzz_forcereplay2.log 2>&1 || { tail -30 gpurun_out/r12i_fuzz_forcereplay2.log; exit 1; }
tail -1 gpurun_out/r12i_fuzz_forcereplay2.log
: > gpurun_out/r12i_time.log
for v in base abl4; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib PQ="4,0.25;3,0.7" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r12i_time.log 2>&1
done
for v in base waves6; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib TRIM=10000 PQ="0.25,0.5;4,2;3,0.7" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r12i_time.log 2>&1
done
grep "G steps" gpurun_out/r12i_time.log
