# GPU call 32: phase statistics of the generic (weighted) exact kernel on cfg 2
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03g
mkdir -p $O
bash scripts/build_stats.sh > $O/build.log 2>&1 || exit 1
WEIGHTS=integer timeout -k 10 300 python scripts/walk_stats_generic.py > $O/stats_integer.log 2>&1 || exit 1
cat $O/stats_integer.log
WEIGHTS=arbitrary timeout -k 10 300 python scripts/walk_stats_generic.py > $O/stats_arbitrary.log 2>&1 || exit 1
cat $O/stats_arbitrary.log
