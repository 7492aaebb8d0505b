set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r13e_time.log
for v in base ablb1 ablb2 abl4; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib PQ="4,0.25" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r13e_time.log 2>&1
done
grep "G steps" gpurun_out/r13e_time.log
