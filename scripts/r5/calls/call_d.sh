# round 5, call d: what the wide path costs (timing-only ablations of the steps on rows >= 65536)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for V in "" wedge_wide1 wedge_wide2; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; fi
  GRAPH=cfg3 TRIM=100000 PQ="0.5,2.0;4.0,0.25;3.0,0.7" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py "abl${V}" 2>&1 | grep "slots\|mode" | tee -a gpurun_out/r7d_time_wide_ablation.log
done
unset N2V_VARIANT_LIB
timeout -k 10 300 python scripts/r5/time_translate.py 2>&1 | grep "cfg4 batch" | tee gpurun_out/r7d_time_translate.log
timeout -k 10 300 python scripts/r5/placement_parts.py 2>&1 | grep "PARTS\|AUDITION\|AGAIN" | tee gpurun_out/r7d_placement_parts.log
