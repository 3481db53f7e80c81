// Wave-per-system MLPG kernels (fast path) -- placeholder until implemented.
#include "common.h"
namespace mlpg {
bool wave_supported(const Problem &, const WinSet &) { return false; }
int launch_wave(hipStream_t, int, int, bool, const Problem &, const WinSet &, int) {
  set_error("wave kernel not built");
  return MLPG_HIP_EINVAL;
}
}  // namespace mlpg
