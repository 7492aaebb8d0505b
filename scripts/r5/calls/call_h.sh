# round 5, call h: weighted lanes, two lane instances (8-slot / 32-slot groups), thresholds
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py -x -q > gpurun_out/r7h_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/r7h_tests_wlanes.log; exit 1; }
tail -2 gpurun_out/r7h_tests_wlanes.log
for T in 256 1024 4096; do
  echo "N2V_WLANES_SHORT=$T" | tee -a gpurun_out/r7h_time_wlanes.log
  N2V_WLANES_SHORT=$T OLD=0 BATCH=471785 KINDS=fp32 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7h_time_wlanes.log
done
echo "N2V_WLANES_SHORT=1024 N2V_WLANES_WAVE_FROM=16384" | tee -a gpurun_out/r7h_time_wlanes.log
N2V_WLANES_SHORT=1024 N2V_WLANES_WAVE_FROM=16384 OLD=0 BATCH=471785 KINDS=fp32 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7h_time_wlanes.log
echo "defaults" | tee -a gpurun_out/r7h_time_wlanes.log
OLD=1 BATCH=47104 KINDS=fp32,fp64 PQ="0.5,2.0;3.0,0.7" timeout -k 10 400 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7h_time_wlanes.log
