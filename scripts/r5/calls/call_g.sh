# round 5, call g: weighted lanes hybrid (thresholds), probe of the biased step shape, second-stage near forms at trim 100000
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py -x -q > gpurun_out/r7g_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/r7g_tests_wlanes.log; exit 1; }
tail -2 gpurun_out/r7g_tests_wlanes.log
for T in 256 1024 4096; do
  echo "N2V_WLANES_MAX_ROW=$T" | tee -a gpurun_out/r7g_time_wlanes.log
  N2V_WLANES_MAX_ROW=$T OLD=0 BATCH=471785 KINDS=fp32 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7g_time_wlanes.log
done
N2V_WLANES_MAX_ROW=1024 OLD=1 BATCH=47104 KINDS=fp32,fp64 PQ="0.5,2.0;3.0,0.7" timeout -k 10 400 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7g_time_wlanes.log
timeout -k 10 300 python scripts/r5/probe_biased.py 2>&1 | grep PROBE | tee gpurun_out/r7g_probe_biased.log
GRAPH=cfg3 TRIM=100000 PQ="3.0,0.7;0.7,3.0" ROUNDS="" timeout -k 10 300 python scripts/r4/time_wedge2.py stage2 2>&1 | grep "slots" | tee gpurun_out/r7g_time_near_stage2.log
