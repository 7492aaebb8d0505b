# variants of n2v_walk_wedge.hip (the one-launch biased kernels) as whole libraries under build_variants/:
#   bash scripts/r4/build_wedge_variants.sh "name:-DFLAG=.. -DFLAG2=.." ...
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/node2vec_amd/csrc
make -s -j8
mkdir -p $R/build_variants
OTHERS=$(ls *.o | grep -v n2v_walk_wedge.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off $flags \
     -c n2v_walk_wedge.hip -o $R/build_variants/wedge_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_variants/libn2v_wedge_$name.so $OTHERS $R/build_variants/wedge_$name.o
  echo built $name
done
