set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_wedge_gpu.py tests/test_walk_gpu.py tests/test_partitioned_gpu.py tests/test_scale_props_gpu.py -x -q > gpurun_out/r7v_tests.log 2>&1 || { tail -40 gpurun_out/r7v_tests.log; exit 1; }
tail -2 gpurun_out/r7v_tests.log
timeout -k 10 300 python scripts/fuzz_walk.py 100 97 2>&1 | tail -2 | tee gpurun_out/r7v_fuzz.log
for V in "" wedge_nointerp; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; else unset N2V_VARIANT_LIB; fi
  GRAPH=cfg3 TRIM=100000 PQ="0.5,2.0;4.0,0.25;4.0,2.0" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py "interp${V:-_on}" 2>&1 | grep "+ slots" | tee -a gpurun_out/r7v_time_interp.log
done
for V in "" wedge_nointerp; do
  if [ -n "$V" ]; then export N2V_VARIANT_LIB=$PWD/build_variants/libn2v_$V.so; else unset N2V_VARIANT_LIB; fi
  GRAPH=cfg4 TRIM=10000 PQ="0.5,2.0;4.0,0.25" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py "interp${V:-_on}" 2>&1 | grep "+ slots" | tee -a gpurun_out/r7v_time_interp.log
done
