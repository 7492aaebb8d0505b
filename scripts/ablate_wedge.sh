# timing-only ablations of the all-tables kernel in the mirror regime (cfg 5, p = 4, q = 0.25)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/node2vec_amd/csrc
mkdir -p ../../build_stats
SRC="n2v_capi.hip n2v_walk.hip n2v_walk_unit.hip n2v_walk_fast.hip n2v_walk_uniform.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip n2v_edge_classes.hip n2v_sync.hip n2v_transform.hip n2v_hops.hip n2v_wedge.hip n2v_walk_wedge.hip"
for a in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off -DN2V_ABLATE_W=$a -shared -o ../../build_stats/libn2v_abl$a.so $SRC &
done
wait
cd $R
GRAPH=cfg5 PQ=4.0,0.25 python scripts/time_wedge_kernel.py "shipped" 2>&1 | grep exact
for a in 1 2 3; do
  N2V_VARIANT_LIB=$R/build_stats/libn2v_abl$a.so GRAPH=cfg5 PQ=4.0,0.25 python scripts/time_wedge_kernel.py "ablation $a" 2>&1 | grep exact
done
