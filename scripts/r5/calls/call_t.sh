set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2 3; do
for V in "" wedge_kmin8 wedge_kmin16; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; else unset N2V_VARIANT_LIB; fi
  GRAPH=cfg4 TRIM=10000 PQ="0.5,2.0;4.0,0.25" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py "k${V:-_min64}" 2>&1 | grep "+ slots" | tee -a gpurun_out/r7x_time_kmin.log
done
done
