# round 6 (second session), call a: the sample index of the wedge lists -- its tests, then cfg 4 trimmed at the
# reference's cap with and without it
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_wedge_index_gpu.py tests/test_wedge_gpu.py -x -q > gpurun_out/r11a_tests_index.log 2>&1 || { tail -40 gpurun_out/r11a_tests_index.log; exit 1; }
tail -3 gpurun_out/r11a_tests_index.log
NOSLOTS=1 timeout -k 10 700 python scripts/r6/time_wedge_index.py r11a > gpurun_out/r11a_time_index_cap100000.log 2>&1 || { tail -30 gpurun_out/r11a_time_index_cap100000.log; exit 1; }
cat gpurun_out/r11a_time_index_cap100000.log
