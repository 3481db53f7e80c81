// constant-coefficient MLPG kernels: backward, double in
#include "mlpg_const_impl.h"
namespace mlpg {
int launch_const_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape) {
  const cst::Plan q = shape == 0 ? cst::make_plan(p, 32, 4) : shape == 2 ? cst::make_plan(p, 16, 8) : shape == 3 ? cst::make_plan(p, 16, 4) : cst::make_plan(p, 16, 2);
  void *sc = scratch(device, st, 4, q.total);
  if (!sc) return MLPG_HIP_ENOMEM;
  if (out_dtype == MLPG_HIP_F32) return cst::launch_t<double, float, true>(st, p, ws, sc, q, true);
  return cst::launch_t<double, double, true>(st, p, ws, sc, q, true);
}
}  // namespace mlpg
