# round 5, call m: weighted walks with table classes in the wave routine; hybrid lanes + waves
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_weighted_lanes_gpu.py tests/test_walk_gpu.py tests/test_edge_cases_gpu.py tests/test_partitioned_gpu.py -x -q > gpurun_out/r7m_tests.log 2>&1 || { tail -40 gpurun_out/r7m_tests.log; exit 1; }
tail -2 gpurun_out/r7m_tests.log
echo "default library (lanes: every row; wave kernel: table classes)" | tee gpurun_out/r7m_time_wlanes.log
OLD=1 BATCH=47104 KINDS=fp32 PQ="0.5,2.0;3.0,0.7" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7m_time_wlanes.log
for T in 512 2048 8192; do
  echo "hybrid: rows of >= $T slots by a wave each" | tee -a gpurun_out/r7m_time_wlanes.log
  N2V_HIP_LIB=$PWD/build_variants/libn2v_wlanes_wave$T.so OLD=0 BATCH=47104 KINDS=fp32 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py 2>&1 | grep "steps/s" | tee -a gpurun_out/r7m_time_wlanes.log
done
