# builds timing-only ablation variants of libn2v_hip.so into build_variants/ (git-ignored)
set -e
cd "$(dirname "$0")/../node2vec_amd/csrc"
mkdir -p ../../build_variants
FLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off"
for A in ${VARIANTS:-1 2 3}; do
  /opt/rocm/bin/hipcc $FLAGS -DN2V_ABLATE=$A -shared -o ../../build_variants/libn2v_a$A.so \
     n2v_capi.hip n2v_walk.hip n2v_walk_fast.hip n2v_alias.hip n2v_sgns.hip n2v_trim.hip &
done
wait
ls -la ../../build_variants
