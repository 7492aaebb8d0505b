# round 6 (second session), call d: why the closed forms decline on long rows (cfg 4 at the reference's cap)
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python scripts/r6/decline_stats.py > gpurun_out/r11d_decline_stats.log 2>&1 || { tail -30 gpurun_out/r11d_decline_stats.log; exit 1; }
cat gpurun_out/r11d_decline_stats.log
