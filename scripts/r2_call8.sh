# GPU call 8: final profiles of the round (kernel trace + PMC passes) on cfg 4 and cfg 3
set -x
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/profile_r2.sh r02h_cfg4 --config cfg4 || exit 1
bash scripts/profile_r2.sh r02h_cfg3 --config cfg3 || exit 1
du -sh $R/gpurun_out
