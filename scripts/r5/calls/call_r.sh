set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for V in big20k big20kd; do
  N2V_VARIANT_LIB=$PWD/build_variants/libn2v_wedge_$V.so timeout -k 10 300 python scripts/r5/big_stats.py 2>&1 | grep "pairings" | tee -a gpurun_out/r7t_big_stats.log
done
for rep in 1 2; do
for V in "" wedge_nokary; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; else unset N2V_VARIANT_LIB; fi
  GRAPH=cfg4 TRIM=10000 PQ="0.5,2.0;4.0,0.25" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py "ab${V:-_kary}" 2>&1 | grep "+ slots" | tee -a gpurun_out/r7t_time_kary_ab.log
done
done
