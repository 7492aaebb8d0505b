# GPU call 12: final state of the round: profiles (trace + PMC) on cfg 4 / cfg 3, the default bench line, end to end on cfg 2
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02x
bash scripts/profile_r2.sh r02x_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02x_cfg3 --config cfg3 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r02x/bench_cfg4.json 2> gpurun_out/r02x/bench_cfg4.err || exit 1
timeout -k 10 300 python bench.py --config cfg2 --cpu-seconds 6 > gpurun_out/r02x/bench_cfg2.json 2> gpurun_out/r02x/bench_cfg2.err || exit 1
timeout -k 10 300 python scripts/e2e_cfg2.py > gpurun_out/r02x/e2e_cfg2.log 2>&1
grep -v amdgpu.ids gpurun_out/r02x/e2e_cfg2.log
du -sh $R/gpurun_out
