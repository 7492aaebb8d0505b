set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7q}
FUZZ_PARTITIONED=1 timeout -k 10 400 python scripts/fuzz_walk.py 200 81 2>&1 | tail -3 | tee gpurun_out/${TAG}_fuzz.log
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 150 82 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_fuzz.log
FUZZ_PQ=rational timeout -k 10 300 python scripts/fuzz_walk.py 150 83 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_fuzz.log
FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 100 84 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_fuzz.log
timeout -k 10 200 python scripts/fuzz_sgns.py 60 85 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_fuzz.log
