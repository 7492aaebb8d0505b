// n2v_sgns_batched.hip -- placeholder until the batched kernel lands (this round): refuses.
#include "n2v_common.h"

extern "C" int n2v_sgns_batched_launch(const int32_t *, int64_t, int32_t, float *, float *,
                                       const uint32_t *, const uint32_t *, const float *,
                                       const n2v_sgns_params *, unsigned long long *, void *) {
  return N2V_EINVAL;
}
