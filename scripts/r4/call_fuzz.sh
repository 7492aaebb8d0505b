set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_multirank_gpu.py -x -q 2>&1 | tail -3
FUZZ_PARTITIONED=1 timeout -k 10 400 python scripts/fuzz_walk.py 240 11 2>&1 | tail -3 | tee gpurun_out/r5j_fuzz_part.log
FUZZ_PARTITIONED=1 FUZZ_PQ=two timeout -k 10 300 python scripts/fuzz_walk.py 120 12 2>&1 | tail -3 | tee -a gpurun_out/r5j_fuzz_part.log
timeout -k 10 300 python scripts/fuzz_walk.py 150 13 2>&1 | tail -3 | tee -a gpurun_out/r5j_fuzz_part.log
