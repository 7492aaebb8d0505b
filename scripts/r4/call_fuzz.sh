set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r5u}
FUZZ_PARTITIONED=1 timeout -k 10 400 python scripts/fuzz_walk.py 240 51 2>&1 | tail -3 | tee gpurun_out/${TAG}_fuzz.log
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 180 52 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_fuzz.log
timeout -k 10 300 python scripts/fuzz_walk.py 180 53 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_fuzz.log
timeout -k 10 200 python scripts/fuzz_sgns.py 90 54 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_fuzz.log
