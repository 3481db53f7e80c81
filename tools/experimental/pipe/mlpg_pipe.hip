// Pipelined strip MLPG kernel: dispatch (the kernel lives in mlpg_pipe_impl.h and is instantiated per dtype in
// mlpg_pipe_{fwd,bwd}_{f32,f64}.hip so that the instantiations compile in parallel).
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace mlpg {

int launch_pipe_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_pipe_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_pipe_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_pipe_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);

namespace {
constexpr int kPipeStripFrames = 48;  // pipe::kC * strip::kM
constexpr int kPipeMaxStrips = 256;   // one workgroup per CU holds two tickets: an utterance's strips must fit the grid
constexpr int kRecBytes = 14 * 64 * 8;
constexpr int kNotResident = -1000;   // = strip::kNotResident
}  // namespace

bool pipe_supported(const Problem &p, const WinSet &ws) {
  if (p.Tmax < 1 || (p.Tmax + kPipeStripFrames - 1) / kPipeStripFrames > kPipeMaxStrips) return false;
  if (ws.nw != 3) return false;  // the streamed level 1 is written for three windows (static, delta, delta-delta)
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return true;
}

int launch_pipe(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws, int device) {
  const int R = (p.Tmax + kPipeStripFrames - 1) / kPipeStripFrames;
  const int ndg = (p.sd + 63) / 64;
  const int dgw = (p.sd + ndg - 1) / ndg;
  const size_t nsg = (size_t)p.B * ndg;
  const size_t ctrl = (((1 + 16 + nsg) * 32 + nsg * (size_t)((R + 31) / 32 * 32)) * sizeof(int) + 255) / 256 * 256;  // >= strip::ctrl_bytes
  unsigned long long gen = 0;
  void *sc = scratch(device, st, 3, ctrl + nsg * R * kRecBytes, &gen);
  if (!sc) return MLPG_HIP_ENOMEM;
  // control words: zero at kernel start; left zero by verdict_kernel (see launch_strip)
  bool zero_ctrl = true;
  {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<unsigned long long, size_t>> clean;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lk(mu);
    auto &c = clean[{device, st}];
    if (!capturing && c.first == gen && ctrl <= c.second) zero_ctrl = false;
    c = {gen, capturing ? (size_t)0 : ctrl};
  }
  // the strip launcher keeps its own bookkeeping of the same scratch slot: after a pipe launch its idea of "clean" may be
  // stale only in the safe direction (verdict_kernel zeroes this launch's whole control area)
  int rc;
  if (!backward)
    rc = dtype == MLPG_HIP_F32 ? launch_pipe_fwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl)
                               : launch_pipe_fwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl);
  else
    rc = dtype == MLPG_HIP_F32 ? launch_pipe_bwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl)
                               : launch_pipe_bwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl);
  if (rc == kNotResident) return launch_strip(st, dtype, out_dtype, backward, p, ws, device);
  return rc;
}

}  // namespace mlpg
