// n2v_trim.hip -- hotspot trimming. Placeholder.
#include "n2v_common.h"
extern "C" int n2v_trim_mark(const int64_t *, int64_t, int64_t, uint64_t, uint8_t *, void *) {
  return N2V_EINVAL;
}
