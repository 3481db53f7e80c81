// constant-coefficient MLPG kernels: forward pass of several streams in one launch (global / unit variances), float
#include "mlpg_const_impl.h"
namespace mlpg {
int launch_const_multi_f32(hipStream_t st, const Problem &p, const WinSet &ws, int device, const StreamMap &sm) {
  const cst::Plan q = cst::make_plan(p, 16, cst::kConstW, sm.total);
  unsigned long long gen = 0;
  void *sc = scratch(device, st, 4, q.total, &gen);
  if (!sc) return MLPG_HIP_ENOMEM;
  const bool fresh = const_scratch_fresh(device, st, gen);
  return cst::launch_multi_t<float>(st, p, ws, sc, q, fresh, device, gen, sm);
}
}  // namespace mlpg
