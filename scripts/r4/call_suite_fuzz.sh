set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5d}
bash scripts/r4/run_suite.sh $TAG
timeout -k 10 200 python scripts/fuzz_walk.py 100 2>&1 | tail -2 | tee gpurun_out/${TAG}_fuzz_walk.log
