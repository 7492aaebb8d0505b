# where the time of the wave kernel with margins goes: timing-only builds (the walks are wrong, the loads real),
# and builds at other occupancies
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r7z}
: > gpurun_out/${TAG}_wm_ablation.log
for v in ${VARIANTS:-sum pass w7 w8}; do
  echo "== build: $v" >> gpurun_out/${TAG}_wm_ablation.log
  N2V_HIP_LIB=$PWD/build_variants/libn2v_wm_$v.so OLD=0 BATCH=471785 timeout -k 10 300 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_wm_ablation.log 2>&1
done
echo "== the product build" >> gpurun_out/${TAG}_wm_ablation.log
OLD=0 BATCH=471785 timeout -k 10 300 python scripts/r5/time_weighted_lanes.py >> gpurun_out/${TAG}_wm_ablation.log 2>&1
grep -v "amdgpu.ids\|per-edge tables" gpurun_out/${TAG}_wm_ablation.log
