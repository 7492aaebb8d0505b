# weighted exact walks: long rows decided with margins by a wave per walker -- tests, then weighted cfg 2
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7x}
timeout -k 10 500 python -m pytest tests/test_weighted_lanes_gpu.py -x -q -m gpu --durations=6 > gpurun_out/${TAG}_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_wlanes.log; exit 1; }
tail -12 gpurun_out/${TAG}_tests_wlanes.log
BOTH=${BOTH:-1} OLD=${OLD:-1} PQ=${PQ:-"0.5,2.0"} BATCH=${BATCH:-47104} timeout -k 10 600 python scripts/r5/time_weighted_lanes.py > gpurun_out/${TAG}_time_wlanes_margins.log 2>&1 || { tail -30 gpurun_out/${TAG}_time_wlanes_margins.log; exit 1; }
cat gpurun_out/${TAG}_time_wlanes_margins.log
