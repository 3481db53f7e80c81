// wave-per-system MLPG: forward, float32 in/out
#include "mlpg_wave_impl.h"
namespace mlpg {
int launch_wave_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device) {
  (void)out_dtype;
  (void)device;
  return launch_t<float, float, false>(st, p, ws);
}
}  // namespace mlpg
