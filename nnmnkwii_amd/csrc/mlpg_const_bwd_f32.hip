// constant-coefficient MLPG kernels: backward, float in
#include "mlpg_const_impl.h"
namespace mlpg {
int launch_const_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape) {
  const cst::Plan q = cst::make_plan(p, 16, cst::kConstW);
  (void)shape;
  unsigned long long gen = 0;
  void *sc = scratch(device, st, 4, q.total, &gen);
  if (!sc) return MLPG_HIP_ENOMEM;
  const bool fresh = const_scratch_fresh(device, st, gen);
  if (out_dtype == MLPG_HIP_F32) return cst::launch_t<float, float, true>(st, p, ws, sc, q, fresh, device, gen);
  return cst::launch_t<float, double, true>(st, p, ws, sc, q, fresh, device, gen);
}
}  // namespace mlpg
