set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r8x}
timeout -k 10 500 python -m pytest tests/test_weighted_lanes_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests_wlanes.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_wlanes.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests_wlanes.log
timeout -k 10 200 python scripts/r5/fuzz_weighted_margins.py 1500 11 2>&1 | grep -v amdgpu.ids
: > gpurun_out/${TAG}_time_wm.log
KINDS=int,fp32 OLD=0 BATCH=47104 PQ="0.5,2.0" timeout -k 10 400 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_time_wm.log 2>&1
grep -v "amdgpu.ids\|per-edge tables" gpurun_out/${TAG}_time_wm.log | cut -c1-250
