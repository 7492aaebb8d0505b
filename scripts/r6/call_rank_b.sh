set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r13g_ablate_cap10000.log
for v in base abl1 abl2 abl4; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib TRIM=10000 PQ="0.5,2;4,0.25;0.25,0.5;3,0.7" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r13g_ablate_cap10000.log 2>&1
done
grep "G steps" gpurun_out/r13g_ablate_cap10000.log
