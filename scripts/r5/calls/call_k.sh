# round 5, call k: weighted lanes with windowed list streams
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py -x -q > gpurun_out/r7k_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/r7k_tests_wlanes.log; exit 1; }
tail -2 gpurun_out/r7k_tests_wlanes.log
for T in 100000000 4096 1024; do
  echo "N2V_WLANES_SHORT=$T" | tee -a gpurun_out/r7k_time_wlanes.log
  N2V_WLANES_SHORT=$T OLD=0 BATCH=47104 KINDS=fp32 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7k_time_wlanes.log
done
