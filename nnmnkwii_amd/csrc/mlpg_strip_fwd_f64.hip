// strip MLPG kernels: forward, double
#include "mlpg_strip_impl.h"
namespace mlpg {
int launch_strip_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl) {
  (void)out_dtype;
  return strip::launch_t<double, double, false>(st, p, ws, scratch, R, ndg, dgw, zero_ctrl);
}
}  // namespace mlpg
