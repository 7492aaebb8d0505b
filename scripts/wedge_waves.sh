# occupancy variants of the wedge kernel (-DN2V_WEDGE_WAVES=4|5|6|8) timed on cfg 4, p = 0.5, q = 2
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/node2vec_amd/csrc
mkdir -p ../../build_stats
SRC="n2v_capi.hip n2v_walk.hip n2v_walk_unit.hip n2v_walk_fast.hip n2v_walk_uniform.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip n2v_edge_classes.hip n2v_sync.hip n2v_transform.hip n2v_hops.hip n2v_wedge.hip n2v_walk_wedge.hip"
for w in 4 5 6 8; do
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off -DN2V_WEDGE_WAVES=$w -shared -o ../../build_stats/libn2v_ww$w.so $SRC &
done
wait
cd $R
for w in 4 5 6 8; do
  N2V_VARIANT_LIB=$R/build_stats/libn2v_ww$w.so python scripts/time_wedge_kernel.py $w
done
