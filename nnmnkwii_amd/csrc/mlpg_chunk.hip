// Launcher of the chunked MLPG kernel (window extents up to 2): see mlpg_chunk_impl.h.
#include <cstdlib>

#include "mlpg_chunk_impl.h"

namespace mlpg {

using namespace chunk;

bool chunk_supported(const Problem &p, const WinSet &ws) {
  if (ws.nw < 1 || ws.nw > kMaxNw || ws.mw < 1 || ws.mw > 2) return false;
  if (p.pitch && p.pitch != p.sd) return false;
  return rows_fit_buffer(p);
}

// the natural-order kernel is the only other one that takes extents of 2; a handful of systems are not worth three launches
bool chunk_preferred(const Problem &p, const WinSet &ws, bool backward) {
  (void)backward;
  return chunk_supported(p, ws) && ws.mw == 2 && (long)p.B * p.sd >= 64 && p.Tmax >= 64;
}

int launch_chunk(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws, int device) {
  if (dtype != out_dtype || !chunk_supported(p, ws)) {
    set_error("MLPG_HIP_ALGO_CHUNK: input dtype = output dtype, 1-3 windows of extent 1 or 2");
    return MLPG_HIP_EINVAL;
  }
  if (dtype == MLPG_HIP_F32) return backward ? launch_chunk_bwd_f32(st, p, ws, device) : launch_chunk_fwd_f32(st, p, ws, device);
  return backward ? launch_chunk_bwd_f64(st, p, ws, device) : launch_chunk_fwd_f64(st, p, ws, device);
}

}  // namespace mlpg
