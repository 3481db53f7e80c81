// chunked MLPG kernels (window extents up to 2): backward, float
#include "mlpg_chunk_impl.h"
namespace mlpg {
int launch_chunk_bwd_f32(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  return chunk::launch_t<float, float, true>(st, p, ws, device);
}
}  // namespace mlpg
