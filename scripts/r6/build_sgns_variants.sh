# variants of the SGNS kernel (n2v_sgns.hip) as whole libraries under build_variants/:  plain = -DN2V_SGNS_COHERENT=0 (the row accesses of rounds 1 - 5)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/node2vec_amd/csrc
make -s -j8
mkdir -p $R/build_variants
OTHERS=$(ls *.o | grep -v "^n2v_sgns.o")
for spec in ${SPECS:-"plain:-DN2V_SGNS_COHERENT=0"}; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off $flags \
     -c n2v_sgns.hip -o $R/build_variants/sgns_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_variants/libn2v_sgns_$name.so $OTHERS $R/build_variants/sgns_$name.o
  echo built $name
done
