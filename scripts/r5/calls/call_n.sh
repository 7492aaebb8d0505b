# round 5, call n: instances <1> and <2> of the slots kernel at 6 / 5 / 4 waves per SIMD
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for V in "" wedge_big5 wedge_big4; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; fi
  GRAPH=cfg4 PQ="4.0,2.0;0.25,0.5;3.0,0.7" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py "waves${V:-_6}" > gpurun_out/r7n_$V.log 2>&1 || { tail -5 gpurun_out/r7n_$V.log; exit 1; }
  grep "+ slots" gpurun_out/r7n_$V.log | tee -a gpurun_out/r7n_time_big_instances.log
done
