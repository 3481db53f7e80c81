// Constant-coefficient MLPG kernels (global / unit variances): dispatch.  The kernels live in mlpg_const_impl.h and
// are instantiated per dtype in mlpg_const_{fwd,bwd}_{f32,f64}.hip so that they compile in parallel.
#include <stdlib.h>
#include "common.h"

namespace mlpg {

int launch_const_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);

namespace {
constexpr int kConstNotResident = -1000;  // = cst::kNotResident (mlpg_const_impl.h)
constexpr int kMaxStrips = 1024;          // strips of one utterance (32 frames each at the small shape)
}  // namespace

// Global (D,) or unit variances, windows of extent <= 1 with at least one dynamic window (mw == 1), two or three
// windows, dense dims (no stream pieces).
bool const_supported(const Problem &p, const WinSet &ws) {
  if (p.var_mode != MLPG_HIP_VAR_GLOBAL && p.var_mode != MLPG_HIP_VAR_UNIT) return false;
  if (ws.nw != 2 && ws.nw != 3) return false;
  if (ws.mw != 1) return false;
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  if (p.pitch && p.pitch != p.sd) return false;
  if (p.Tmax < 1 || p.sd < 1 || p.B < 1) return false;
  if ((p.Tmax + 31) / 32 > kMaxStrips) return false;
  return true;
}

// AUTO: lanes = static dims, so narrow streams (lf0: 1 dim, bap: 5) stay with the wave-per-system kernel.
bool const_preferred(const Problem &p, const WinSet &ws) {
  if (!const_supported(p, ws)) return false;
  const int ndg = (p.sd + 63) / 64, dgw = (p.sd + ndg - 1) / ndg;
  return dgw >= 16;
}

int launch_const(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                 int device) {
  // shape 0: 32-frame chunks, 4 per strip (128-frame strips); shape 1: 16-frame chunks, 2 per strip -- when the
  // launch has too few 128-frame strips to fill the machine
  const int ndg = (p.sd + 63) / 64;
  const long big_items = (long)p.B * ndg * ((p.Tmax + 127) / 128);
  int shape = big_items >= 1024 ? 0 : 1;
  if (const char *e = getenv("MLPG_CONST_SHAPE")) shape = atoi(e);  // experiments: 0 = 32 x 4, 1 = 16 x 2, 2 = 16 x 8
  int rc;
  if (!backward)
    rc = dtype == MLPG_HIP_F32 ? launch_const_fwd_f32(st, out_dtype, p, ws, device, shape)
                               : launch_const_fwd_f64(st, out_dtype, p, ws, device, shape);
  else
    rc = dtype == MLPG_HIP_F32 ? launch_const_bwd_f32(st, out_dtype, p, ws, device, shape)
                               : launch_const_bwd_f64(st, out_dtype, p, ws, device, shape);
  if (rc == kConstNotResident) {
    // fewer workgroups can be resident than an utterance has strips: nothing was enqueued
    return wave_supported(p, ws) ? launch_wave(st, dtype, out_dtype, backward, p, ws, device)
                                 : launch_generic(st, dtype, out_dtype, backward, p, ws, device);
  }
  return rc;
}

}  // namespace mlpg
