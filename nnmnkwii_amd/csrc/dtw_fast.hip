// fastdtw kernel -- placeholder until implemented.
#include "common.h"
namespace mlpg {
int launch_fastdtw(hipStream_t, int, const double *, const double *, const int32_t *, const int32_t *, int, int, int,
                   int, int, int32_t *, int32_t *, int32_t *, double *) {
  set_error("fastdtw kernel not built");
  return MLPG_HIP_ERUNTIME;
}
}  // namespace mlpg
