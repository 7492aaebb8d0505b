set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for V in "" wedge_all5; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; fi
  GRAPH=cfg4 PQ="0.5,2.0;4.0,0.25;2.0,1.0" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py "waves${V:-_default}" > gpurun_out/r7o_$V.log 2>&1 || { tail -5 gpurun_out/r7o_$V.log; exit 1; }
  grep "+ slots" gpurun_out/r7o_$V.log | tee -a gpurun_out/r7o_time_instances03.log
done
