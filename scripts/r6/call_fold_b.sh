# round 6 (second session): the tree with folded slots against the commit before it, same box, cfg 4 trimmed at 10 000
# (no wide row there: the difference is what the list type of the closed forms costs the ordinary rows)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r12b_ab_cap10000.log
for rep in 1 2; do
for v in new prev; do
  lib=$PWD/node2vec_amd/libn2v_hip.so
  [ $v = prev ] && lib=$PWD/build_variants/libn2v_prev.so
  N2V_HIP_LIB=$lib TRIM=10000 PQ="0.5,2;4,0.25;3,0.7;0.25,0.5;4,2" REPS=4 timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r12b_ab_cap10000.log 2>&1 || { tail -30 gpurun_out/r12b_ab_cap10000.log; exit 1; }
done
done
grep "G steps" gpurun_out/r12b_ab_cap10000.log
