# diagnostic build with per-wave counters and cycle stamps (-DN2V_STATS) into build_stats/ (git-ignored);
# scripts/walk_stats.py loads it by path (the product loader has no override)
set -e
cd "$(dirname "$0")/../node2vec_amd/csrc"
mkdir -p ../../build_stats
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off -DN2V_STATS \
  -shared -o ../../build_stats/libn2v_stats.so n2v_capi.hip n2v_walk.hip n2v_walk_unit.hip n2v_walk_fast.hip \
  n2v_walk_uniform.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip n2v_edge_classes.hip n2v_sync.hip n2v_transform.hip n2v_hops.hip n2v_wedge.hip n2v_walk_wedge.hip
ls -la ../../build_stats
