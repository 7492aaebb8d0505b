// n2v_alias.hip -- K1 first-order alias tables. Placeholder.
#include "n2v_common.h"
extern "C" int n2v_alias_build(const int64_t *, const float *, int64_t, int32_t *, double *,
                               uint32_t *, void *) {
  return N2V_EINVAL;
}
