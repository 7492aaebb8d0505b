# instruction counts of the wave margin kernel by phase: PMC passes over the timing-only builds
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r8z}
for v in sum pass; do
  N2V_HIP_LIB=$PWD/build_variants/libn2v_wm_$v.so PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" bash scripts/r5/pmc_wm.sh ${TAG}_$v | grep -A5 "margin_kernel<float, true, false>"
done
