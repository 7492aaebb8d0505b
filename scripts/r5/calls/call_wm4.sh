# weighted exact walks with margins: other (p, q), fp64 weights, small batches against the wave kernel
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7zf}
: > gpurun_out/${TAG}_time_wm_more.log
OLD=1 BATCH=4710 BIG=4710 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_time_wm_more.log 2>&1
OLD=1 BATCH=47104 BIG=47104 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_time_wm_more.log 2>&1
OLD=0 BATCH=471785 PQ="4.0,0.25;3.0,0.7;1.0,2.0" timeout -k 10 600 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_time_wm_more.log 2>&1
KINDS=fp64 OLD=0 BATCH=471785 PQ="0.5,2.0" timeout -k 10 300 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_time_wm_more.log 2>&1
grep -v "amdgpu.ids" gpurun_out/${TAG}_time_wm_more.log
