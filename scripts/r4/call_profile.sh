set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4zz}
timeout -k 10 1100 bash scripts/r4/profile_r4.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { tail -30 gpurun_out/${TAG}_profile.log; exit 1; }
tail -5 gpurun_out/${TAG}_profile.log
