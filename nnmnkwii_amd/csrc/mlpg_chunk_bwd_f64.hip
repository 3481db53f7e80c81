// chunked MLPG kernels (window extents up to 2): backward, double
#include "mlpg_chunk_impl.h"
namespace mlpg {
int launch_chunk_bwd_f64(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  return chunk::launch_t<double, double, true>(st, p, ws, device);
}
}  // namespace mlpg
