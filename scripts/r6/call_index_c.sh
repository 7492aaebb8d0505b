# round 6 (second session), call c: where the time of the exact biased kernels goes on cfg 4 trimmed at the
# reference's cap -- timing-only ablation builds (-DN2V_ABLATE_STEP=k, n2v_wedge_step.h)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/r11c_ablate_cap100000.log
for v in base abl1 abl2 abl3 abl4 abl5 abl6 abl7 ablw1; do
  lib=$PWD/build_variants/libn2v_wedge_$v.so
  [ $v = base ] && lib=$PWD/node2vec_amd/libn2v_hip.so
  pq="0.5,2;4,0.25;3,0.7"
  [ $v = base ] && reps=2 || reps=3
  N2V_HIP_LIB=$lib PQ="$pq" REPS=$reps timeout -k 10 300 python scripts/r6/time_variant.py $v >> gpurun_out/r11c_ablate_cap100000.log 2>&1 || { tail -30 gpurun_out/r11c_ablate_cap100000.log; exit 1; }
done
grep "G steps" gpurun_out/r11c_ablate_cap100000.log
