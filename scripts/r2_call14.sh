# GPU call 12: final state of the round: profiles (trace + PMC) on cfg 4 / cfg 3, the default bench line, end to end on cfg 2
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r02n
GRAPH=cfg5 PQ=4.0,0.25 python scripts/time_wedge_kernel.py "cfg5 4,0.25 early offset" > gpurun_out/r02n_time_cfg5.log 2>&1; grep exact gpurun_out/r02n_time_cfg5.log
bash scripts/profile_r2.sh r02n_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02n_cfg3 --config cfg3 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r02n/bench_cfg4.json 2> gpurun_out/r02n/bench_cfg4.err || exit 1
timeout -k 10 300 python bench.py --config cfg2 --cpu-seconds 6 > gpurun_out/r02n/bench_cfg2.json 2> gpurun_out/r02n/bench_cfg2.err || exit 1
timeout -k 10 300 python scripts/e2e_cfg2.py > gpurun_out/r02n/e2e_cfg2.log 2>&1
grep -v amdgpu.ids gpurun_out/r02n/e2e_cfg2.log
du -sh $R/gpurun_out
