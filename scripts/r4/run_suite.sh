# the whole -m gpu suite as the driver runs it (+ durations), then the biased timing
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r4j}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${TAG}_tests_gpu.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_gpu.log; exit 1; }
tail -18 gpurun_out/${TAG}_tests_gpu.log
GRAPH=cfg4 PQ="0.5,2;4,0.25;4,2" ROUNDS="" timeout -k 10 400 python scripts/r4/time_wedge2.py $TAG > gpurun_out/${TAG}_time_cfg4.log 2>&1 || { tail -20 gpurun_out/${TAG}_time_cfg4.log; exit 1; }
grep -v amdgpu gpurun_out/${TAG}_time_cfg4.log
