# partitioned walking with capacity-bounded mailboxes: its tests, then cfg 2 in 8 parts (one-process form, the
# ranks' form bounded and with exact sizes)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7w}
timeout -k 10 600 python -m pytest tests/test_partitioned_gpu.py tests/test_multirank_gpu.py -x -q -m gpu --durations=8 > gpurun_out/${TAG}_tests_part.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_part.log; exit 1; }
tail -14 gpurun_out/${TAG}_tests_part.log
SECTIONS=${SECTIONS:-} PQ="1,1;0.5,2;4,0.25" FORWARD=${FORWARD:-} timeout -k 10 400 python scripts/r4/time_partitioned.py > gpurun_out/${TAG}_time_partitioned.log 2>&1 || { tail -30 gpurun_out/${TAG}_time_partitioned.log; exit 1; }
cat gpurun_out/${TAG}_time_partitioned.log
