// Wave-per-system MLPG kernels: dispatch (the kernels live in mlpg_wave_impl.h and are
// instantiated per dtype in mlpg_wave_{fwd,bwd}_{f32,f64}.hip so that they compile in parallel).
#include "common.h"

namespace mlpg {

int launch_wave_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device);
int launch_wave_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device);
int launch_wave_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws);
int launch_wave_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws);

bool wave_supported(const Problem &p, const WinSet &ws) {
  if (p.Tmax > 64 * 32 || p.Tmax < 1) return false;
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return true;
}

int launch_wave(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                int device) {
  if (!backward) return dtype == MLPG_HIP_F32 ? launch_wave_fwd_f32(st, out_dtype, p, ws, device) : launch_wave_fwd_f64(st, out_dtype, p, ws, device);
  return dtype == MLPG_HIP_F32 ? launch_wave_bwd_f32(st, out_dtype, p, ws) : launch_wave_bwd_f64(st, out_dtype, p, ws);
}

}  // namespace mlpg
