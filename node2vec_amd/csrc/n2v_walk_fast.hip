// n2v_walk_fast.hip -- K2 fast mode (first-order alias tables + rejection). Placeholder.
#include "n2v_common.h"
extern "C" int n2v_walk_fast_launch(const n2v_graph *, const int32_t *, int64_t, int32_t, int32_t,
                                    double, double, uint64_t, int32_t *, uint8_t *, uint32_t *,
                                    void *) {
  return N2V_EINVAL;
}
