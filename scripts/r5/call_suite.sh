# the whole -m gpu suite as the driver runs it (+ durations), smoke, then the default bench
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7l}
timeout -k 10 1050 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_tests_gpu.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests_gpu.log; exit 1; }
tail -22 gpurun_out/${TAG}_tests_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
