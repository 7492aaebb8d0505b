# the library with -DN2V_DEFER_HOP=${DEFER:-0} (every biased step gathers the hop entry of `pick` first: rounds 4 - 5), for A/B runs
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/node2vec_amd/csrc
make -s -j8
mkdir -p $R/build_variants
OBJ=""
for f in $(grep -l "n2v_wedge_step.h" *.hip); do
  b=${f%.hip}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -ffp-contract=off -DN2V_DEFER_HOP=${DEFER:-0} \
     -c $f -o $R/build_variants/${b}_defer${DEFER:-0}.o
  OBJ="$OBJ $R/build_variants/${b}_defer${DEFER:-0}.o"
  SKIP="$SKIP|^$b.o"
done
OTHERS=$(ls *.o | grep -v -E "${SKIP#|}")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build_variants/libn2v_defer${DEFER:-0}.so $OTHERS $OBJ
echo built defer${DEFER:-0}
