// Strip MLPG kernels: dispatch (the kernels live in mlpg_strip_impl.h and are instantiated per
// dtype in mlpg_strip_{fwd,bwd}_{f32,f64}.hip so that they compile in parallel).
#include <map>
#include <mutex>
#include <utility>
#include <limits.h>
#include <string.h>

#include "common.h"

namespace mlpg {

int launch_strip_fwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_strip_fwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_strip_bwd_f64(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_strip_bwd_f32(hipStream_t st, int out_dtype, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl);
int launch_strip_multi_f64(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl, const StreamMap &sm);
int launch_strip_multi_f32(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, int ndg, int dgw, bool zero_ctrl, const StreamMap &sm);
int launch_strip_tr_f64(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, bool zero_ctrl, const StreamMap &sm);
int launch_strip_tr_f32(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch, int R, bool zero_ctrl, const StreamMap &sm);

namespace {
#ifndef MLPG_STRIP_W
#define MLPG_STRIP_W 4
#endif
#ifndef MLPG_STRIP_M
#define MLPG_STRIP_M 16
#endif
constexpr int kStripFrames = MLPG_STRIP_M * MLPG_STRIP_W;   // strip::kW * strip::kM
constexpr int kMaxStrips = 256;    // strips of one utterance must be able to be resident together (2 per CU)
constexpr int kRecBytes = 14 * 64 * 8;
constexpr int kStripNotResident = kStripMultiNotResident;  // = strip::kNotResident (mlpg_strip_impl.h)
}  // namespace

bool strip_supported(const Problem &p, const WinSet &ws) {
  if (p.Tmax < 1 || (p.Tmax + kStripFrames - 1) / kStripFrames > kMaxStrips) return false;
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return rows_fit_buffer(p);
}

// AUTO policy (measured on MI355X with tools/algo_sweep.py, profiles/r02_algo_sweep.txt): the strip kernel puts
// static dims on lanes and the strips of one utterance on different CUs; the wave-per-system kernel keeps a whole
// utterance in one workgroup.  The strip kernel wins when its lanes are filled (dims per 64-lane group >= 48: 60, 64,
// 128 static dims; not 25 or 80 = 2 x 40) and the launch has enough 64-frame strips to occupy the persistent grid
// (>= 512: two per CU); with per-frame variances that is 0.233 vs 0.285 ms forward (float32: 0.168 vs 0.239 ms)
// on the config-2 shape, 0.032 vs 0.075 ms at 256 x 100 frames.  Beyond 1024 frames the wave kernel needs 32 frames per lane
// (register spills, one workgroup per CU) or does not apply at all (T > 2048), and the strip kernel takes over
// for every stream of >= 16 dims.
bool strip_preferred(const Problem &p, const WinSet &ws, bool backward, int in_dtype) {
  if (!strip_supported(p, ws)) return false;
  if (p.sd >= 16 && p.Tmax > 1024) return true;
  const int ndg = (p.sd + 63) / 64, dgw = (p.sd + ndg - 1) / ndg;
  const long nitems = (long)p.B * ndg * ((p.Tmax + kStripFrames - 1) / kStripFrames);
  if (dgw < 48) return false;
  // global (D,) variances: 0.172 ms (strip) vs 0.196 ms (wave) when the windows hold, but 0.213 ms when some static
  // dim's delta variance is several times tighter than its static one (then every strip of the group sweeps the
  // whole utterance) -- the usual case for variances taken from data statistics; unit variances tie (0.169 vs 0.177 ms
  // float64, 0.137 vs 0.128 ms float32).  Both stay with the wave kernel.
  if (p.var_mode != MLPG_HIP_VAR_FRAME) return false;
  // backward (config-2 shape): 0.296 (strip) vs 0.397 ms (wave) in float64, 0.273 vs 0.314 ms in float32; 64 x 500:
  // 0.044 vs 0.053 ms -- the same rule both ways
  (void)backward;
  (void)in_dtype;
  return nitems >= 512;
}

// Transposed form (round 5): a NARROW stream -- 1 .. 32 static dims: lf0, bap, vuv of a Merlin-style row, or the piece a merged launch
// left over -- leaves most of the strip kernel's 64 lanes idle, and the wave-per-system kernel that took such streams walks each
// (utterance, dim) system with one wavefront (512 x 2000 frames of lf0: 0.08 ms for 57 MB).  Here the lanes of a group run over
// 64 / sd consecutive UTTERANCES x the stream's dims (StreamMap::tr_u): same kernel, same records, only the lane's columns carry the
// utterance's offset.  Forward, three windows of extent <= 1 (what the MULTI instantiation is compiled for), any variance mode, with or
// without a lengths vector (a group runs to its longest utterance, a lane's own dead frames enter by per-lane selects); every offset
// inside the 2 GB window of the group's first utterance.
bool strip_tr_supported(const Problem &p, const WinSet &ws, bool backward, int in_dtype, int out_dtype) {
  if (backward || in_dtype != out_dtype || ws.nw != 3) return false;
  if (p.sd < 1 || p.sd > 32 || p.B < 2 || !strip_supported(p, ws)) return false;
  const int u = 64 / p.sd;
  const long ld = p.ld_in > p.ld_out ? p.ld_in : p.ld_out;
  return (double)u * (double)p.Tmax * (double)ld * 8.0 < 2147483647.0;
}
// AUTO takes it where it beat the wave-per-system kernel (tools/dbg/narrow_time.py, profiles/r05_tr_narrow.txt): full lane groups,
// and enough (group, strip) items that the persistent grid's fixed costs -- about 30 us: one launch, one round of items, the verdict
// launch -- are paid back: 256 items in float64, 512 in float32 (the wave kernel moves half the bytes there); beyond 1024 frames
// the wave kernel needs 32 frames per lane and 64 items suffice (512 x 2000 x 5 dims: 0.087 against 0.188 ms; 512 x 2000 x 1:
// 0.059 against 0.074; 256 x 1000 x 25: 0.108 against 0.152; but 256 x 1000 x 1, 64 items: 0.030 against 0.020).
bool strip_tr_preferred(const Problem &p, const WinSet &ws, bool backward, int in_dtype, int out_dtype) {
  if (!strip_tr_supported(p, ws, backward, in_dtype, out_dtype)) return false;
  const int u = 64 / p.sd;
  const long items = (long)((p.B + u - 1) / u) * ((p.Tmax + kStripFrames - 1) / kStripFrames);
  if (p.B < u) return false;
  if (p.Tmax > 1024) return items >= 64;
  return items >= (in_dtype == MLPG_HIP_F32 ? 512 : 256);
}

namespace {
std::mutex g_clean_mu;
std::map<std::pair<int, hipStream_t>, std::pair<unsigned long long, size_t>> g_clean;  // allocation, zero bytes at its head
// a launch that enqueued nothing (kStripNotResident) after strip_scratch: nothing is known about the control area
void strip_scratch_forget(hipStream_t st, int device) {
  std::lock_guard<std::mutex> lk(g_clean_mu);
  g_clean[{device, st}].second = 0;
}
// scratch (control words + records) of one launch and whether its control area is known to be zero already
void *strip_scratch(hipStream_t st, int device, size_t nsg, int R, bool *zero_ctrl) {
  const size_t ctrl = (((1 + 16 + nsg) * 32 + nsg * (size_t)((R + 31) / 32 * 32)) * sizeof(int) + 255) / 256 * 256;  // >= strip::ctrl_bytes
  unsigned long long gen = 0;
  void *sc = scratch(device, st, 3, ctrl + nsg * R * kRecBytes, &gen);
  if (!sc) return nullptr;
  // The control words must be zero when the kernel starts.  verdict_kernel leaves them zero again, so a launch on
  // the same scratch whose control area is not larger than the previous one's (everything beyond it held records)
  // needs no memset -- one dependent launch less per call.  Never trusted while the stream is being captured
  // into a graph: a replay may follow launches this bookkeeping has not seen.
  *zero_ctrl = true;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lk(g_clean_mu);
  auto &c = g_clean[{device, st}];
  if (!capturing && c.first == gen && ctrl <= c.second) *zero_ctrl = false;
  c = {gen, capturing ? (size_t)0 : ctrl};
  return sc;
}
}  // namespace

// ---- XCC_ID probe ----
// Long utterances (more strips than a work list may hold) are dealt to all eight per-XCD lists in blocks, and a strip that
// waits for its whole utterance then needs EVERY list to be drawn by somebody.  A workgroup draws from the list of the XCD
// it runs on and helps the others only once its own is exhausted, so on a device whose workgroups do not land on eight XCDs
// in roughly equal numbers (CPX / DPX / QPX partitions, parts with fewer XCDs) some lists would have no home workgroups and
// the waiting ones would spin on strips nobody draws.  Asked of the hardware itself, once per device: a grid of small
// workgroups, each adding one to the counter of its HW_REG_XCC_ID.
namespace {
__global__ void xcc_probe_kernel(int *hist) {  // hist: pinned host memory (system-scope atomics)
  if (threadIdx.x == 0) atomicAdd_system(hist + (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7), 1);
}
}  // namespace
bool strip_xcd_lists_ok(hipStream_t st) {
  static std::mutex mu;
  static std::map<int, bool> seen;
  static std::map<int, int> failed;  // probes that could not run (per device): after three the answer is "no" for good
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(mu);
  auto it = seen.find(dev);
  if (it != seen.end()) return it->second;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;  // (not cached: the first launch outside a capture probes)
  }
  // The probe runs inside an asynchronous launch path (the first long utterance on a device), so it must not synchronise the device
  // or disturb anybody's stream capture: its own non-blocking stream, counters in pinned host memory that the
  // kernel updates itself (no device allocation, no blocking copy, nothing on the NULL stream), a wait on that stream only; and for
  // its duration this thread's calls are in relaxed capture mode, so that another thread's global-mode capture is not invalidated.
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool mode_set = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  bool ok = false, probed = false;
  int *h = nullptr;
  hipStream_t ps = nullptr;
  constexpr int kBlocks = 4096;
  if (hipHostMalloc((void **)&h, 8 * sizeof(int), hipHostMallocDefault) == hipSuccess) {
    for (int k = 0; k < 8; ++k) h[k] = 0;
    if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) == hipSuccess) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(kBlocks), dim3(64), 0, ps, h);
      if (hipGetLastError() == hipSuccess && hipStreamSynchronize(ps) == hipSuccess) {
        int lo = h[0], hi = h[0];
        for (int k = 1; k < 8; ++k) { lo = h[k] < lo ? h[k] : lo; hi = h[k] > hi ? h[k] : hi; }
        ok = lo > 0 && 2 * lo >= hi;
        probed = true;
      }
      (void)hipStreamDestroy(ps);
    }
    (void)hipHostFree(h);
  }
  (void)hipGetLastError();
  if (mode_set) (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (probed) seen[dev] = ok;
  else if (++failed[dev] >= 3) seen[dev] = false;  // (a runtime that keeps refusing the probe: stop asking on every launch)
  return ok;
}

int launch_strip_tr(hipStream_t st, int dtype, const Problem &p, const WinSet &ws, int device) {
  const int R = (p.Tmax + kStripFrames - 1) / kStripFrames;
  StreamMap sm;
  memset(&sm, 0, sizeof(sm));
  for (int q = 0; q < 4; ++q) sm.begin[q] = INT_MAX;
  sm.n = 1;
  sm.begin[0] = 0;
  sm.sd[0] = p.pitch ? p.pitch : p.sd;  // the window pitch (a piece of a stream: the whole stream's static dim)
  sm.tr_nd = p.sd;
  sm.tr_u = 64 / p.sd;
  sm.total = sm.tr_u * sm.tr_nd;
  sm.tr_B = p.B;
  sm.tr_in = (int)((long)p.Tmax * p.ld_in);
  sm.tr_out = (int)((long)p.Tmax * p.ld_out);
  sm.tr_stat = p.ld_status;
  const size_t nsg = (size_t)(p.B + sm.tr_u - 1) / sm.tr_u;
  bool zero_ctrl = true;
  void *sc = strip_scratch(st, device, nsg, R, &zero_ctrl);
  if (!sc) return MLPG_HIP_ENOMEM;
  const int rc = dtype == MLPG_HIP_F32 ? launch_strip_tr_f32(st, p, ws, sc, R, zero_ctrl, sm) : launch_strip_tr_f64(st, p, ws, sc, R, zero_ctrl, sm);
  if (rc == kStripNotResident) strip_scratch_forget(st, device);
  return rc;
}

int launch_strip(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                 int device, bool try_tr) {
  // a narrow stream asked for by name: the transposed form where it applies (the plain form would run 64 / sd times the items);
  // try_tr = false: the caller has just been told that the grid cannot hold that form
  if (try_tr && strip_tr_supported(p, ws, backward, dtype, out_dtype)) {
    const int rc = launch_strip_tr(st, dtype, p, ws, device);
    if (rc != kStripNotResident) return rc;
  }
  const int R = (p.Tmax + kStripFrames - 1) / kStripFrames;
  const int ndg = (p.sd + 63) / 64;
  const int dgw = (p.sd + ndg - 1) / ndg;
  const size_t nsg = (size_t)p.B * ndg;
  bool zero_ctrl = true;
  void *sc = strip_scratch(st, device, nsg, R, &zero_ctrl);
  if (!sc) return MLPG_HIP_ENOMEM;
  int rc;
  if (!backward)
    rc = dtype == MLPG_HIP_F32 ? launch_strip_fwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl)
                               : launch_strip_fwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl);
  else
    rc = dtype == MLPG_HIP_F32 ? launch_strip_bwd_f32(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl)
                               : launch_strip_bwd_f64(st, out_dtype, p, ws, sc, R, ndg, dgw, zero_ctrl);
  if (rc == kStripNotResident) {
    strip_scratch_forget(st, device);
    // fewer workgroups can be resident than an utterance has strips (a smaller device, or an occupancy the runtime
    // reports lower than expected): nothing was enqueued; the natural-order kernel has no such requirement
    return launch_generic(st, dtype, out_dtype, backward, p, ws, device);
  }
  return rc;
}

// Several streams of one batch in one launch (mlpg_hip_forward_streams): the lanes run over the static dims of all of
// them, groups of 64.  Returns kStripNotResident (nothing enqueued) if the grid cannot hold an utterance: the caller
// then runs the streams one by one.
int launch_strip_multi(hipStream_t st, int dtype, const Problem &p, const WinSet &ws, const StreamMap &sm, int device) {
  const int R = (p.Tmax + kStripFrames - 1) / kStripFrames;
  const int ndg = (sm.total + 63) / 64;
  const int dgw = 64;  // full groups first: 66 dims = 64 + 2, not 33 + 33 (a group costs its frames whatever its lanes)
  const size_t nsg = (size_t)p.B * ndg;
  bool zero_ctrl = true;
  void *sc = strip_scratch(st, device, nsg, R, &zero_ctrl);
  if (!sc) return MLPG_HIP_ENOMEM;
  const int rc = dtype == MLPG_HIP_F32 ? launch_strip_multi_f32(st, p, ws, sc, R, ndg, dgw, zero_ctrl, sm)
                                       : launch_strip_multi_f64(st, p, ws, sc, R, ndg, dgw, zero_ctrl, sm);
  if (rc == kStripNotResident) strip_scratch_forget(st, device);
  return rc;
}

}  // namespace mlpg
