# round 6 (second session): where the time goes with folded slots in place, cfg 4 trimmed at 100 000
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r12c_ablate_fold_cap100000.log
for v in base abl2 abl3 abl4 abl6 abl7; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  N2V_HIP_LIB=$lib PQ="3,0.7;4,0.25;0.5,2" REPS=3 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r12c_ablate_fold_cap100000.log 2>&1 || { tail -30 gpurun_out/r12c_ablate_fold_cap100000.log; exit 1; }
done
grep "G steps" gpurun_out/r12c_ablate_fold_cap100000.log
