# round 5, call e: what the wide path costs (timing-only ablations of the steps on rows >= 65536)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for V in wedge_wide1 wedge_wide2; do
  export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so
  GRAPH=cfg3 TRIM=100000 PQ="0.5,2.0;4.0,0.25;3.0,0.7" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py "abl${V}" > gpurun_out/r7e_$V.log 2>&1 || { tail -5 gpurun_out/r7e_$V.log; exit 1; }
  grep "slots\|mode" gpurun_out/r7e_$V.log | tee -a gpurun_out/r7e_time_wide_ablation.log
done
