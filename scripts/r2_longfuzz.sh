set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02t
timeout -k 10 520 python scripts/fuzz_walk.py 480 20261003 > gpurun_out/r02t/fuzz_walk_long.log 2>&1
tail -1 gpurun_out/r02t/fuzz_walk_long.log
FUZZ_PQ=extreme timeout -k 10 280 python scripts/fuzz_walk.py 240 20261004 > gpurun_out/r02t/fuzz_walk_extreme_long.log 2>&1
tail -1 gpurun_out/r02t/fuzz_walk_extreme_long.log
timeout -k 10 200 python scripts/fuzz_sgns.py 150 20261005 > gpurun_out/r02t/fuzz_sgns_long.log 2>&1
tail -1 gpurun_out/r02t/fuzz_sgns_long.log
