// n2v_sgns.hip -- K3 SGNS. Placeholder.
#include "n2v_common.h"
extern "C" int n2v_sgns_train(const int32_t *, int64_t, int32_t, float *, float *,
                              const uint32_t *, const n2v_sgns_params *, unsigned long long *,
                              void *) {
  return N2V_EINVAL;
}
