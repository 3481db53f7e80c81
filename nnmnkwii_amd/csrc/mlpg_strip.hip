// Strip MLPG kernels: dispatch (the kernels live in mlpg_strip_impl.h and are instantiated per
// dtype in mlpg_strip_{fwd,bwd}_{f32,f64}.hip so that they compile in parallel).
#include "common.h"

namespace mlpg {

int launch_strip_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw);
int launch_strip_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw);
int launch_strip_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw);
int launch_strip_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw);

namespace {
constexpr int kStripFrames = 64;   // strip::kW * strip::kM
constexpr int kMaxStrips = 256;    // strips of one utterance must be able to be resident together (2 per CU)
constexpr int kRecBytes = 14 * 64 * 8;
}  // namespace

bool strip_supported(const Problem &p, const WinSet &ws) {
  if (p.Tmax < 1 || (p.Tmax + kStripFrames - 1) / kStripFrames > kMaxStrips) return false;
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return true;
}

// AUTO policy (measured on MI355X, profiles/r02_notes.md): the strip kernel puts static dims on lanes and strips of
// one utterance on different CUs; its inter-workgroup level costs ~10 us per 64-frame strip, which the
// wave-per-system kernel (whole utterance in one workgroup) does not pay.  At T <= 1024 the two tie on wide streams
// with per-frame variances and the wave kernel wins where the traffic is lighter (global / unit variances,
// backward); beyond that the wave kernel needs 32 frames per lane (register spills, one workgroup per CU) or does
// not apply at all (T > 2048), and the strip kernel takes over.
bool strip_preferred(const Problem &p, const WinSet &ws) { return strip_supported(p, ws) && p.sd >= 16 && p.Tmax > 1024; }

int launch_strip(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                 int device) {
  const int R = (p.Tmax + kStripFrames - 1) / kStripFrames;
  const int ndg = (p.sd + 63) / 64;
  const int dgw = (p.sd + ndg - 1) / ndg;
  const size_t nsg = (size_t)p.B * ndg;
  const size_t ctrl = (((1 + 8 + nsg) * 32 + nsg * (size_t)((R + 31) / 32 * 32)) * sizeof(int) + 255) / 256 * 256;  // strip::ctrl_bytes
  void *sc = scratch(device, st, 3, ctrl + nsg * R * kRecBytes);
  if (!sc) return MLPG_HIP_ENOMEM;
  if (!backward)
    return dtype == MLPG_HIP_F32 ? launch_strip_fwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw)
                                 : launch_strip_fwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw);
  return dtype == MLPG_HIP_F32 ? launch_strip_bwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw)
                               : launch_strip_bwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw);
}

}  // namespace mlpg
