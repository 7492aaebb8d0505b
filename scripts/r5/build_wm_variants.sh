# variants of the wave kernel with margins (n2v_walk_wlanes.hip) as whole libraries under build_variants/:
#   timing-only builds (sum, pass: -DN2V_WM_ABLATE) and waves per SIMD (w7, w8: -DN2V_WM_WAVES_PER_SIMD)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/node2vec_amd/csrc
make -s -j8
mkdir -p $R/build_variants
OTHERS=$(ls *.o | grep -v n2v_walk_wlanes.o)
for spec in ${SPECS:-"sum:-DN2V_WM_ABLATE=1" "pass:-DN2V_WM_ABLATE=2" "w7:-DN2V_WM_WAVES_PER_SIMD=7" "w8:-DN2V_WM_WAVES_PER_SIMD=8"}; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off $flags \
     -c n2v_walk_wlanes.hip -o $R/build_variants/wlanes_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_variants/libn2v_wm_$name.so $OTHERS $R/build_variants/wlanes_$name.o
  echo built $name
done
