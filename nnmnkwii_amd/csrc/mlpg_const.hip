// Constant-coefficient MLPG kernels (global / unit variances): dispatch.  The kernels live in mlpg_const_impl.h and
// are instantiated per dtype in mlpg_const_{fwd,bwd}_{f32,f64}.hip so that they compile in parallel.
#include <algorithm>
#include <map>
#include <vector>
#include <mutex>
#include <utility>
#include "common.h"

namespace mlpg {

int launch_const_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, int device, int shape);
int launch_const_multi_f64(hipStream_t st, const Problem &p, const WinSet &ws, int device, const StreamMap &sm);
int launch_const_multi_f32(hipStream_t st, const Problem &p, const WinSet &ws, int device, const StreamMap &sm);

bool const_scratch_fresh(int device, hipStream_t stream, unsigned long long gen) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, unsigned long long> seen;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  // never trusted while the stream is being captured into a graph: a replay may follow launches this has not seen
  const bool capturing = hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lk(mu);
  unsigned long long &g = seen[{device, stream}];
  const bool fresh = capturing || g != gen;
  g = capturing ? 0ull : gen;
  return fresh;
}

bool const_unit_table_cached(int device, hipStream_t stream, unsigned long long gen, bool fresh, const double *key, int n) {
  struct Entry {
    unsigned long long gen = 0;
    std::vector<double> key;
    bool captured = false;  // a capture ran on this (device, stream): its graph holds THIS stream's scratch table and rewrites
                            // it (fresh = 1) on every replay, on whatever stream and without passing through here -- the host
                            // can no longer know what the table holds, so the device-side key check runs on every launch again
  };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Entry> seen;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lk(mu);
  Entry &e = seen[{device, stream}];
  // (the table is shared with the global-variance launches of this stream: any of those in between rewrites it, and the
  // device-side key would notice -- the host-side one cannot, so a global-variance launch forgets the entry, see below)
  if (capturing) e.captured = true;
  const bool same = !fresh && !capturing && !e.captured && e.gen == gen && (int)e.key.size() == n && std::equal(e.key.begin(), e.key.end(), key);
  e.gen = capturing ? 0ull : gen;
  if (n > 0) e.key.assign(key, key + n);
  else e.key.clear();
  if (capturing) e.key.clear();
  return same;
}

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
  return rows_fit_buffer(p);
}

// AUTO (measured on MI355X, tools/dbg/const_time.py): one workgroup walks one (utterance, dim group) sequence, so the
// launch needs about one sequence per CU to fill the chip (256 x 1000 x 60 float64: 0.148 ms against 0.205 ms for the
// wave-per-system kernel and 0.190 ms for the strip kernel; 512 x 2000 x 60: 0.46 against 0.94 / 0.70 ms), and lanes are
// static dims, so narrow streams (lf0: 1 dim, bap: 5) and small batches (64 x 500 x 60: 0.050 against 0.041 ms) stay
// with the wave-per-system kernel.
bool const_preferred(const Problem &p, const WinSet &ws) {
  if (!const_supported(p, ws)) return false;
  const int ndg = (p.sd + 63) / 64, dgw = (p.sd + ndg - 1) / ndg;
  return dgw >= 32 && (long)p.B * ndg >= 192;
}

int launch_const(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                 int device) {
  const int shape = 0;  // 16-frame chunks, 8 per super-step, one workgroup per CU (mlpg_const_impl.h: launch_t)
  if (!backward)
    return dtype == MLPG_HIP_F32 ? launch_const_fwd_f32(st, out_dtype, p, ws, device, shape)
                                 : launch_const_fwd_f64(st, out_dtype, p, ws, device, shape);
  return dtype == MLPG_HIP_F32 ? launch_const_bwd_f32(st, out_dtype, p, ws, device, shape)
                               : launch_const_bwd_f64(st, out_dtype, p, ws, device, shape);
}


// Several streams of one batch in one launch (mlpg_hip_forward_streams with global (D,) or unit variances): the lanes run over
// the static dims of all of them, full groups of 64 (the caller packs the streams).  p.sd = the merged dims, p.mean / p.var /
// p.out = the parent arrays (column 0), three windows of extent <= 1.
int launch_const_multi(hipStream_t st, int dtype, const Problem &p, const WinSet &ws, const StreamMap &sm, int device) {
  return dtype == MLPG_HIP_F32 ? launch_const_multi_f32(st, p, ws, device, sm) : launch_const_multi_f64(st, p, ws, device, sm);
}

}  // namespace mlpg
