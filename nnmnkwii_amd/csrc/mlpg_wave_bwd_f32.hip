// wave-per-system MLPG: backward, float32 inputs
#include "mlpg_wave_impl.h"
namespace mlpg {
int launch_wave_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws) {
  (void)out_dtype;
  return out_dtype == MLPG_HIP_F32 ? launch_t<float, float, true>(st, p, ws)
                                   : launch_t<float, double, true>(st, p, ws);
}
}  // namespace mlpg
