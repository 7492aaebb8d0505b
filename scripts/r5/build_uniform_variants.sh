# timing-only variants of n2v_walk_uniform.hip as whole libraries under build_variants/ (placement diagnosis):
#   uniform_nostore (no path stores), uniform_noread (no table reads)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/node2vec_amd/csrc
make -s -j8
mkdir -p $R/build_variants
OTHERS=$(ls *.o | grep -v n2v_walk_uniform.o)
for spec in "nostore:-DN2V_ABLATE_UNIFORM=1" "noread:-DN2V_ABLATE_UNIFORM=2"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off $flags \
     -c n2v_walk_uniform.hip -o $R/build_variants/uniform_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_variants/libn2v_uniform_$name.so $OTHERS $R/build_variants/uniform_$name.o
  echo built $name
done
