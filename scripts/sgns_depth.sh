# SGNS lookahead depth 2 (dim <= 128) against the shipped depth 1, at 10^8 and 10^6 rows
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/node2vec_amd/csrc
mkdir -p ../../build_stats
SRC="n2v_capi.hip n2v_walk.hip n2v_walk_unit.hip n2v_walk_fast.hip n2v_walk_uniform.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip n2v_edge_classes.hip n2v_sync.hip n2v_transform.hip n2v_hops.hip n2v_wedge.hip n2v_walk_wedge.hip"
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off '-DN2V_SGNS_DEPTH(VV)=((VV)<=2?2:((VV)<=8?1:0))' -shared -o ../../build_stats/libn2v_depth2.so $SRC
cd $R
for n in 1e8 1e6; do
  python scripts/time_sgns_scale.py $n 128 2>&1 | grep Mpairs
  N2V_VARIANT_LIB=$R/build_stats/libn2v_depth2.so python scripts/time_sgns_scale.py $n 128 2>&1 | grep Mpairs
done
