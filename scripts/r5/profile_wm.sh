# kernel trace of weighted exact walks (lanes + margins), weighted cfg 2, all walkers
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7y}
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_prof
OLD=0 BATCH=471785 PQ=${PQ:-"0.5,2.0"} timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o wm -- python3 scripts/r5/time_weighted_lanes.py > gpurun_out/${TAG}_prof.log 2>&1 || { tail -30 gpurun_out/${TAG}_prof.log; exit 1; }
tail -5 gpurun_out/${TAG}_prof.log
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_wm_kernel_stats.csv
head -14 gpurun_out/${TAG}_wm_kernel_stats.csv | cut -c1-220
rm -rf gpurun_out/${TAG}_prof
