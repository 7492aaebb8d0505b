set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7z}
timeout -k 10 500 python -m pytest tests/test_weighted_lanes_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_wlanes.log; exit 1; }
tail -3 gpurun_out/${TAG}_tests_wlanes.log
bash scripts/r5/call_wm_ablate.sh $TAG
