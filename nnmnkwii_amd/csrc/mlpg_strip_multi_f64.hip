// strip MLPG kernels: forward pass of several streams in one launch, double
#include "mlpg_strip_impl.h"
namespace mlpg {
int launch_strip_multi_f64(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl, const StreamMap &sm) {
  return strip::launch_multi_t<double, double>(st, p, ws, scratch, R, ndg, dgw, zero_ctrl, sm);
}
// ... and the transposed form: one narrow stream, the lanes over several utterances (StreamMap::tr_u)
int launch_strip_tr_f64(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, bool zero_ctrl, const StreamMap &sm) {
  return strip::launch_tr_t<double, double>(st, p, ws, scratch, R, zero_ctrl, sm);
}
}  // namespace mlpg
