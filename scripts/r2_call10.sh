# GPU call 10: fuzz with extreme p, q; profiles (kernel trace + PMC passes) of the closed-form build on cfg 4 / cfg 3;
# cfg 2 bench line for the record
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02j
FUZZ_PQ=extreme timeout -k 10 300 python scripts/fuzz_walk.py 150 2718 > gpurun_out/r02j/fuzz_walk_extreme.log 2>&1
tail -2 gpurun_out/r02j/fuzz_walk_extreme.log
grep -q "fuzz ok" gpurun_out/r02j/fuzz_walk_extreme.log || exit 1
bash scripts/profile_r2.sh r02j_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02j_cfg3 --config cfg3 || exit 1
timeout -k 10 400 python bench.py --config cfg2 --cpu-seconds 6 > gpurun_out/r02j/bench_cfg2.json 2> gpurun_out/r02j/bench_cfg2.err || exit 1
timeout -k 10 400 python bench.py --config cfg3 --cpu-seconds 6 > gpurun_out/r02j/bench_cfg3.json 2> gpurun_out/r02j/bench_cfg3.err || exit 1
du -sh $R/gpurun_out
