// chunked MLPG kernels (window extents up to 2): forward, double
#include "mlpg_chunk_impl.h"
namespace mlpg {
int launch_chunk_fwd_f64(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  return chunk::launch_t<double, double, false>(st, p, ws, device);
}
}  // namespace mlpg
