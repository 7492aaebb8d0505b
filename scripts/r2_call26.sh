set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02z
timeout -k 10 300 python -m pytest tests/test_scale_cfg345_gpu.py -x -q -k cfg3 > gpurun_out/r02z/test_cfg3.log 2>&1
tail -2 gpurun_out/r02z/test_cfg3.log
for seed in 31 32 33; do
  timeout -k 10 300 python scripts/fuzz_walk.py 200 $seed > gpurun_out/r02z/fuzz_walk_$seed.log 2>&1
  tail -1 gpurun_out/r02z/fuzz_walk_$seed.log
  grep -q "fuzz ok" gpurun_out/r02z/fuzz_walk_$seed.log || exit 1
done
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 200 34 > gpurun_out/r02z/fuzz_walk_extreme_34.log 2>&1
tail -1 gpurun_out/r02z/fuzz_walk_extreme_34.log
timeout -k 10 300 python scripts/fuzz_sgns.py 150 35 > gpurun_out/r02z/fuzz_sgns_35.log 2>&1
tail -1 gpurun_out/r02z/fuzz_sgns_35.log
