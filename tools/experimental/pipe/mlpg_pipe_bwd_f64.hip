// pipelined strip MLPG kernel: backward, double gradients in, float32 or float64 out
#include "mlpg_pipe_impl.h"
namespace mlpg {
int launch_pipe_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl) {
  if (out_dtype == MLPG_HIP_F32) return pipe::launch_t<double, float, true>(st, p, ws, scratch, R, ndg, dgw, zero_ctrl);
  return pipe::launch_t<double, double, true>(st, p, ws, scratch, R, ndg, dgw, zero_ctrl);
}
}  // namespace mlpg
