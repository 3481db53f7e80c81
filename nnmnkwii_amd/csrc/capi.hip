// C-ABI entry points of libmlpg_hip.so (declared in include/mlpg_hip.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include <limits.h>
#include <string.h>

#include "common.h"

namespace mlpg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

std::atomic<long long> g_launches[kCountKinds];
void note_launch(int kind) {
  if (kind >= 0 && kind < kCountKinds) g_launches[kind].fetch_add(1);
}

// ---- grow-only scratch, one set of buffers per (device, stream) ------------------------------
// Launches on different streams of one device never share a buffer; a buffer is only ever
// reused, grown or freed behind work of its own stream.
namespace {
constexpr int kMaxDevices = 16, kSlots = 7;
struct Slot {
  void *ptr = nullptr;
  size_t bytes = 0;
  unsigned long long gen = 0;  // allocation number: a caller that remembers the buffer's content checks this
};
unsigned long long g_scratch_gen = 0;
struct StreamScratch {
  Slot slot[kSlots];
};
std::map<std::pair<int, hipStream_t>, StreamScratch> g_scratch;
std::mutex g_mu;
}  // namespace

void *scratch(int device, hipStream_t stream, int slot, size_t bytes, unsigned long long *gen) {
  if (device < 0 || device >= kMaxDevices || slot < 0 || slot >= kSlots) {
    set_error("scratch: bad device/slot %d/%d", device, slot);
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  Slot &s = g_scratch[std::make_pair(device, stream)].slot[slot];
  if (s.bytes >= bytes && s.ptr) {
    if (gen) *gen = s.gen;
    return s.ptr;
  }
  if (s.ptr) {
    // the previous users of this buffer were enqueued on this very stream
    (void)hipStreamSynchronize(stream);
    (void)hipFree(s.ptr);
    s.ptr = nullptr;
    s.bytes = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  void *p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    want = bytes;
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      set_error("scratch: hipMalloc of %zu bytes failed", bytes);
      return nullptr;
    }
  }
  s.ptr = p;
  s.bytes = want;
  s.gen = ++g_scratch_gen;
  if (gen) *gen = s.gen;
  return p;
}

namespace {

// Side streams for the independent streams of one multi-stream call (per device, created once).
struct SideStreams {
  static constexpr int kN = 3;
  hipStream_t st[kN];
  hipEvent_t fork, join[kN];
  std::mutex mu;  // held by mlpg_hip_forward_streams while it enqueues (two host threads on one device)
};
SideStreams *g_side[kMaxDevices] = {};

SideStreams *side_streams(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_side[device]) return g_side[device];
  SideStreams *s = new SideStreams();
  bool ok = hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) == hipSuccess;
  for (int q = 0; ok && q < SideStreams::kN; ++q)
    ok = hipStreamCreateWithFlags(&s->st[q], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&s->join[q], hipEventDisableTiming) == hipSuccess;
  if (!ok) {  // no side streams: the caller falls back to its own stream
    (void)hipGetLastError();
    delete s;
    return nullptr;
  }
  g_side[device] = s;
  return s;
}

int pack_windows(int nw, const int32_t *wl, const int32_t *wu, const double *wc, WinSet *ws) {
  if (nw < 1 || nw > kMaxWindows || !wl || !wu || !wc) {
    set_error("num_windows must be in [1, %d] and the window tables non-NULL", kMaxWindows);
    return MLPG_HIP_EINVAL;
  }
  memset(ws, 0, sizeof(*ws));
  ws->nw = nw;
  int off = 0;
  for (int w = 0; w < nw; ++w) {
    if (wl[w] < 0 || wu[w] < 0 || wl[w] > kMaxExtent || wu[w] > kMaxExtent) {
      set_error("window %d: extents (l=%d, u=%d) must be in [0, %d]", w, wl[w], wu[w], kMaxExtent);
      return MLPG_HIP_EINVAL;
    }
    const int n = wl[w] + wu[w] + 1;
    if (off + n > kMaxCoef) {
      set_error("too many window coefficients (> %d)", kMaxCoef);
      return MLPG_HIP_EINVAL;
    }
    ws->l[w] = wl[w];
    ws->u[w] = wu[w];
    ws->off[w] = off;
    for (int i = 0; i < n; ++i) ws->c[off + i] = wc[off + i];
    off += n;
    if (wl[w] + wu[w] > ws->q) ws->q = wl[w] + wu[w];
    const int m = wl[w] > wu[w] ? wl[w] : wu[w];
    if (m > ws->mw) ws->mw = m;
  }
  return 0;
}

}  // namespace
int pack_windows_public(int nw, const int32_t *wl, const int32_t *wu, const double *wc, WinSet *ws) {
  return pack_windows(nw, wl, wu, wc, ws);
}
namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (device != prev && hipSetDevice(device) != hipSuccess) ok = false;
    target = device;
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
  }
  int target = -1;
};

int check_common(int B, int Tmax, int D, int nw) {
  if (B < 0 || Tmax < 0 || D < 0 || nw < 1) {
    set_error("negative size (B=%d, Tmax=%d, D=%d) or num_windows=%d", B, Tmax, D, nw);
    return MLPG_HIP_EINVAL;
  }
  if (D % nw != 0) {
    set_error("D=%d is not a multiple of num_windows=%d", D, nw);
    return MLPG_HIP_EINVAL;
  }
  return 0;
}

}  // namespace

// Kernel choice, in ONE place (round 6): which kernel a problem goes to, as a pure function of the problem -- used by
// dispatch_solve, which launches it (and asks again with a kernel excluded when that kernel declines at launch time: the FIR form
// whose tap table fails its decay test, the transposed strip form the grid cannot hold), and by mlpg_hip_forward_streams, which must
// know beforehand whether a stream will occupy the device with a persistent grid (stream_takes_tr).
// AUTO = the FIR form for float32 unit variances without lengths; the strip kernel's transposed form for narrow streams; the
// constant-coefficient kernel for global / unit variances; the chunked kernel for window extents of 2; the strip kernel for wide
// streams (static dims on lanes); the wave-per-system kernel for narrow ones and small launches; else the natural-order kernel.
enum Route { kRouteFir, kRouteStripTr, kRouteConst, kRouteChunk, kRouteStrip, kRouteWave, kRouteGeneric, kRoutePieceBackward };
enum { kExcludeFir = 1, kExcludeTr = 2 };
static Route route_of(int in_dtype, int out_dtype, int algo, bool backward, const Problem &p, const WinSet &ws, unsigned exclude) {
  const bool piece = p.pitch && p.pitch != p.sd;  // some of a stream's static dims: window pitch != number of dims
  if (algo == MLPG_HIP_ALGO_PIPE) algo = strip_supported(p, ws) ? MLPG_HIP_ALGO_STRIP : MLPG_HIP_ALGO_AUTO;  // retired: the scheme it pipelined
  if (algo == MLPG_HIP_ALGO_FIR) return kRouteFir;
  // (the FIR form first: it has no chain at all; not for a piece of a stream)
  if (algo == MLPG_HIP_ALGO_AUTO && !piece && !(exclude & kExcludeFir) && fir_shape_supported(p, ws, in_dtype, out_dtype) && fir_preferred(p, backward))
    return kRouteFir;
  // a narrow stream (or the piece a merged launch left over): the strip kernel with its lanes over several utterances
  if (!(exclude & kExcludeTr) && ((algo == MLPG_HIP_ALGO_AUTO && strip_tr_preferred(p, ws, backward, in_dtype, out_dtype)) ||
                                 (algo == MLPG_HIP_ALGO_STRIP && piece && strip_tr_supported(p, ws, backward, in_dtype, out_dtype))))
    return kRouteStripTr;
  if (piece) {  // otherwise the kernels that take the pitch separately
    if (backward) return kRoutePieceBackward;
    return wave_supported(p, ws) ? kRouteWave : kRouteGeneric;
  }
  if (algo == MLPG_HIP_ALGO_AUTO) {
    if (const_preferred(p, ws)) return kRouteConst;
    if (in_dtype == out_dtype && chunk_preferred(p, ws, backward)) return kRouteChunk;
    if (strip_preferred(p, ws, backward, in_dtype) || (strip_supported(p, ws) && !wave_supported(p, ws))) return kRouteStrip;
    return wave_supported(p, ws) ? kRouteWave : kRouteGeneric;
  }
  switch (algo) {
    case MLPG_HIP_ALGO_CONST: return kRouteConst;
    case MLPG_HIP_ALGO_CHUNK: return kRouteChunk;
    case MLPG_HIP_ALGO_STRIP: return kRouteStrip;
    case MLPG_HIP_ALGO_WAVE: return kRouteWave;
    default: return kRouteGeneric;
  }
}

int dispatch_solve(hipStream_t st, int in_dtype, int out_dtype, int algo, bool backward, const Problem &p,
                   const WinSet &ws, int device) {
  if (algo < MLPG_HIP_ALGO_AUTO || algo > MLPG_HIP_ALGO_FIR) {
    set_error("unknown algo %d", algo);
    return MLPG_HIP_EINVAL;
  }
  if (algo == MLPG_HIP_ALGO_WAVE && !wave_supported(p, ws)) {
    set_error("MLPG_HIP_ALGO_WAVE does not support this problem (T=%d, half-bandwidth %d)", p.Tmax, ws.q);
    return MLPG_HIP_EINVAL;
  }
  if (algo == MLPG_HIP_ALGO_STRIP && !strip_supported(p, ws)) {
    set_error("MLPG_HIP_ALGO_STRIP does not support this problem (T=%d, half-bandwidth %d)", p.Tmax, ws.q);
    return MLPG_HIP_EINVAL;
  }
  if (algo == MLPG_HIP_ALGO_CONST && !const_supported(p, ws)) {
    set_error("MLPG_HIP_ALGO_CONST needs global or unit variances and 2-3 windows of extent <= 1 (var_mode %d, %d windows, T=%d)", p.var_mode, ws.nw, p.Tmax);
    return MLPG_HIP_EINVAL;
  }
  if (algo == MLPG_HIP_ALGO_CHUNK && (in_dtype != out_dtype || !chunk_supported(p, ws))) {
    set_error("MLPG_HIP_ALGO_CHUNK: input dtype = output dtype, 1-3 windows of extent 1 or 2 (%d windows, extent %d)", ws.nw, ws.mw);
    return MLPG_HIP_EINVAL;
  }
  if (algo == MLPG_HIP_ALGO_FIR && !fir_shape_supported(p, ws, in_dtype, out_dtype)) {
    set_error("MLPG_HIP_ALGO_FIR: unit variances, float32 in and out, no lengths, T >= 96, 1-3 windows of extent <= 2 (the first a single tap)");
    return MLPG_HIP_EINVAL;
  }
  unsigned exclude = 0;
  for (;;) {
    switch (route_of(in_dtype, out_dtype, algo, backward, p, ws, exclude)) {
      case kRouteFir: {
        const int rc = launch_fir(st, backward, p, ws, device);
        if (rc != kFirNotApplicable) return rc;
        if (algo == MLPG_HIP_ALGO_FIR) {
          set_error("MLPG_HIP_ALGO_FIR: the inverse of this window set does not decay to 2^-26 within 24 frames (or the stream is being captured before the tap table exists)");
          return MLPG_HIP_EINVAL;
        }
        exclude |= kExcludeFir;  // the other kernels take the call
        continue;
      }
      case kRouteStripTr: {
        const int rc = launch_strip_tr(st, in_dtype, p, ws, device);
        if (rc != kStripMultiNotResident) return rc;
        exclude |= kExcludeTr;  // a launch the grid cannot hold: the other kernels
        continue;
      }
      case kRoutePieceBackward:
        set_error("a stream piece (pitch %d, %d dims) has no backward pass", p.pitch, p.sd);
        return MLPG_HIP_EINVAL;
      case kRouteConst: return launch_const(st, in_dtype, out_dtype, backward, p, ws, device);
      case kRouteChunk: return launch_chunk(st, in_dtype, out_dtype, backward, p, ws, device);
      // (the strip launcher tries the transposed form itself when asked for by name: not again after it declined here)
      case kRouteStrip: return launch_strip(st, in_dtype, out_dtype, backward, p, ws, device, !(exclude & kExcludeTr));
      case kRouteWave: return launch_wave(st, in_dtype, out_dtype, backward, p, ws, device);
      default: return launch_generic(st, in_dtype, out_dtype, backward, p, ws, device);
    }
  }
}

namespace {

int solve_entry(int device, void *stream, int in_dtype, int out_dtype, int algo, bool backward, const void *mean,
                const void *var, int var_mode, const void *grad_out, const int32_t *lengths, int B, int Tmax,
                int D, int nw, const int32_t *wl, const int32_t *wu, const double *wc, void *out,
                int32_t *status) {
  if (int rc = check_common(B, Tmax, D, nw)) return rc;
  if ((in_dtype != MLPG_HIP_F32 && in_dtype != MLPG_HIP_F64) ||
      (out_dtype != MLPG_HIP_F32 && out_dtype != MLPG_HIP_F64)) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  if (var_mode < 0 || var_mode > 2 || (var_mode != MLPG_HIP_VAR_UNIT && !var)) {
    set_error("bad var_mode %d / NULL var", var_mode);
    return MLPG_HIP_EINVAL;
  }
  if (B * (long)Tmax * D > 0 && (!out || (backward ? !grad_out : !mean))) {
    set_error("NULL data pointer");
    return MLPG_HIP_EINVAL;
  }
  WinSet ws;
  if (int rc = pack_windows(nw, wl, wu, wc, &ws)) return rc;
  if (B == 0 || Tmax == 0 || D == 0) return 0;
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  Problem p;
  p.mean = mean;
  p.var = var;
  p.grad_out = grad_out;
  p.lengths = lengths;
  p.out = out;
  p.status = status;
  p.var_mode = var_mode;
  p.B = B;
  p.Tmax = Tmax;
  p.D = D;
  p.sd = D / nw;
  p.ld_in = D;
  p.ld_gout = backward ? D / nw : 0;
  p.ld_out = backward ? D : D / nw;
  p.ld_status = D / nw;
  hipStream_t st = (hipStream_t)stream;
  return dispatch_solve(st, in_dtype, out_dtype, algo, backward, p, ws, device);
}

// One stream of a multi-stream batch: a column slice [in_col, in_col + nw*sd) of the (B, Tmax, ld_in)
// parameter matrices, trajectory written to columns [out_col, out_col + sd) of the (B, Tmax, ld_out) output.
// [d_first, d_first + d_count): the stream's static dims the problem holds (all of them by default).
int stream_problem(int dtype, const void *mean, const void *var, int var_mode, long ld_in, const int32_t *lengths, int B, int Tmax,
                   const mlpg_hip_stream_t &sm, const int32_t *wl, const int32_t *wu, const double *wc, void *out, long ld_out,
                   int32_t *status, int ld_status, int status_col, int d_first, int d_count, Problem *pp, WinSet *ws) {
  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8;
  const int nw = sm.num_windows;
  const int sd = d_count < 0 ? sm.static_dim : d_count;
  status_col += d_first;
  size_t coff = 0;
  for (int w = 0; w < sm.win_first; ++w) coff += (size_t)(wl[w] + wu[w] + 1);
  if (nw > 0)
    if (int rc = pack_windows(nw, wl + sm.win_first, wu + sm.win_first, wc + coff, ws)) return rc;
  Problem &p = *pp;
  p.mean = (const char *)mean + esz * (size_t)(sm.in_col + d_first);
  p.var = var ? (const char *)var + esz * (size_t)(sm.in_col + d_first) : nullptr;
  p.grad_out = nullptr;
  p.lengths = lengths;
  p.out = (char *)out + esz * (size_t)(sm.out_col + d_first);
  p.status = status ? status + status_col : nullptr;
  p.var_mode = var_mode;
  p.B = B;
  p.Tmax = Tmax;
  p.D = nw * sd;
  p.sd = sd;
  p.pitch = sd != sm.static_dim ? sm.static_dim : 0;
  p.ld_in = ld_in;
  p.ld_gout = 0;
  p.ld_out = ld_out;
  p.ld_status = ld_status;
  return 0;
}

int stream_entry(int device, hipStream_t st, int dtype, int algo, const void *mean, const void *var, int var_mode,
                 long ld_in, const int32_t *lengths, int B, int Tmax, const mlpg_hip_stream_t &sm, const int32_t *wl,
                 const int32_t *wu, const double *wc, void *out, long ld_out, int32_t *status, int ld_status,
                 int status_col, int d_first = 0, int d_count = -1) {
  Problem p;
  WinSet ws;
  if (int rc = stream_problem(dtype, mean, var, var_mode, ld_in, lengths, B, Tmax, sm, wl, wu, wc, out, ld_out, status, ld_status,
                              status_col, d_first, d_count, &p, &ws))
    return rc;
  if (sm.num_windows == 0) return launch_copy_cols(st, dtype, p.mean, ld_in, lengths, B, Tmax, p.sd, p.out, ld_out);
  // a piece of a stream (what a merged launch left over) goes to the kernels that take the window pitch separately, whatever
  // kernel was asked for the call as a whole
  if (p.pitch && (algo == MLPG_HIP_ALGO_CONST || algo == MLPG_HIP_ALGO_CHUNK || algo == MLPG_HIP_ALGO_FIR)) algo = MLPG_HIP_ALGO_AUTO;
  return dispatch_solve(st, dtype, dtype, algo, false, p, ws, device);
}

// Will stream_entry hand this stream (or piece) to the transposed strip form?  Such a launch is a persistent grid like the merged
// launch's: mlpg_hip_forward_streams queues it on the caller's stream behind that one instead of beside it on a side stream (two
// persistent grids that share the device each hold fewer workgroups than their work lists were dealt for).
bool stream_takes_tr(int dtype, int algo, const void *mean, const void *var, int var_mode, long ld_in, const int32_t *lengths, int B,
                     int Tmax, const mlpg_hip_stream_t &sm, const int32_t *wl, const int32_t *wu, const double *wc, void *out,
                     long ld_out, int d_first = 0, int d_count = -1) {
  if (sm.num_windows != 3) return false;
  Problem p;
  WinSet ws;
  if (stream_problem(dtype, mean, var, var_mode, ld_in, lengths, B, Tmax, sm, wl, wu, wc, out, ld_out, nullptr, 1, 0, d_first, d_count, &p, &ws))
    return false;
  if (p.pitch && (algo == MLPG_HIP_ALGO_CONST || algo == MLPG_HIP_ALGO_CHUNK || algo == MLPG_HIP_ALGO_FIR)) algo = MLPG_HIP_ALGO_AUTO;
  return route_of(dtype, dtype, algo, false, p, ws, 0) == kRouteStripTr;  // (the decision dispatch_solve will take)
}

}  // namespace
long long host_chunks_on_device(int device);  // host_api.hip
}  // namespace mlpg

using namespace mlpg;

extern "C" {

__attribute__((visibility("default"))) int mlpg_hip_abi_version(void) { return 14; }

__attribute__((visibility("default"))) long long mlpg_hip_launch_count(int kind) {
  if (kind >= 100) return host_chunks_on_device(kind - 100);  // chunks the host-memory calls enqueued on device kind - 100
  return kind >= 0 && kind < kCountKinds ? g_launches[kind].load() : -1;
}

__attribute__((visibility("default"))) const char *mlpg_hip_last_error(void) { return g_err; }

__attribute__((visibility("default"))) int mlpg_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

__attribute__((visibility("default"))) void mlpg_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (auto &kv : g_scratch) {
    (void)hipSetDevice(kv.first.first);
    (void)hipDeviceSynchronize();
    for (int s = 0; s < kSlots; ++s)
      if (kv.second.slot[s].ptr) (void)hipFree(kv.second.slot[s].ptr);
  }
  g_scratch.clear();
  for (int d = 0; d < kMaxDevices; ++d) {
    if (!g_side[d]) continue;
    (void)hipSetDevice(d);
    for (int q = 0; q < SideStreams::kN; ++q) {
      (void)hipStreamSynchronize(g_side[d]->st[q]);
      (void)hipStreamDestroy(g_side[d]->st[q]);
      (void)hipEventDestroy(g_side[d]->join[q]);
    }
    (void)hipEventDestroy(g_side[d]->fork);
    delete g_side[d];
    g_side[d] = nullptr;
  }
  fir_shutdown();
  host_api_shutdown();
  if (prev >= 0) (void)hipSetDevice(prev);
}

__attribute__((visibility("default"))) int mlpg_hip_forward(int device, void *stream, int dtype, int algo,
                                                            const void *mean, const void *var, int var_mode,
                                                            const int32_t *lengths, int B, int Tmax, int D,
                                                            int num_windows, const int32_t *win_l_h,
                                                            const int32_t *win_u_h, const double *win_coef_h,
                                                            void *out, int32_t *status) {
  return solve_entry(device, stream, dtype, dtype, algo, false, mean, var, var_mode, nullptr, lengths, B, Tmax, D,
                     num_windows, win_l_h, win_u_h, win_coef_h, out, status);
}

__attribute__((visibility("default"))) int mlpg_hip_forward_streams(
    int device, void *stream, int dtype, int algo, const void *mean, const void *var, int var_mode, int64_t ld_in,
    const int32_t *lengths, int B, int Tmax, int num_streams, const mlpg_hip_stream_t *streams_h, int total_windows,
    const int32_t *win_l_h, const int32_t *win_u_h, const double *win_coef_h, void *out, int64_t ld_out,
    int32_t *status) {
  if (B < 0 || Tmax < 0 || num_streams < 0 || total_windows < 0 || ld_in < 0 || ld_out < 0 ||
      ld_in > INT32_MAX || ld_out > INT32_MAX) {
    set_error("forward_streams: negative or oversized size argument");
    return MLPG_HIP_EINVAL;
  }
  if (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  if (num_streams > 0 && !streams_h) {
    set_error("forward_streams: NULL stream table");
    return MLPG_HIP_EINVAL;
  }
  long sd_total = 0;
  for (int k = 0; k < num_streams; ++k) {
    const mlpg_hip_stream_t &sm = streams_h[k];
    const long width = (long)(sm.num_windows > 0 ? sm.num_windows : 1) * sm.static_dim;
    if (sm.static_dim < 0 || sm.num_windows < 0 || sm.in_col < 0 || sm.out_col < 0 || sm.win_first < 0 ||
        sm.win_first + sm.num_windows > total_windows || sm.in_col + width > ld_in ||
        (long)sm.out_col + sm.static_dim > ld_out) {
      set_error("forward_streams: stream %d does not fit (in_col=%d, out_col=%d, static_dim=%d, num_windows=%d)", k,
                sm.in_col, sm.out_col, sm.static_dim, sm.num_windows);
      return MLPG_HIP_EINVAL;
    }
    sd_total += sm.static_dim;
  }
  if (var_mode < 0 || var_mode > 2 || (var_mode != MLPG_HIP_VAR_UNIT && !var)) {
    set_error("bad var_mode %d / NULL var", var_mode);
    return MLPG_HIP_EINVAL;
  }
  if (total_windows > 0 && (!win_l_h || !win_u_h || !win_coef_h)) {
    set_error("forward_streams: NULL window tables");
    return MLPG_HIP_EINVAL;
  }
  if (B == 0 || Tmax == 0 || sd_total == 0) return 0;
  if (!mean || !out) {
    set_error("NULL data pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  // The streams are independent: the widest one runs on the caller's stream, every other one on a side stream of
  // this device that is forked from and joined back into the caller's stream with events (no host synchronisation,
  // capturable), so that a narrow stream's launch fills the tail of the wide one instead of queueing behind it.
  if (num_streams > 64) {
    set_error("forward_streams: more than 64 streams");
    return MLPG_HIP_EINVAL;
  }
  int status_cols[64];
  {
    int c = 0;
    for (int k = 0; k < num_streams; ++k) { status_cols[k] = c; c += streams_h[k].static_dim; }
  }
  // Streams with the same three windows (extent <= 1) and per-frame variances can share ONE strip-kernel launch: their
  // static dims sit side by side on the lanes (66 = 60 + 1 + 5 dims of a Merlin-style row: one group of 64 and one of
  // 2), every row of the batch is fetched once instead of once per stream.  Taken when it does not add lane groups
  // and the launch is one the strip kernel would be chosen for anyway.
  bool merged_flag[64] = {};
  // The merged launch holds full groups of 64 lanes only: a last group of a few lanes would hold its workgroup slots
  // as long as a full group's while moving almost nothing (66 dims in one launch: 1.39 ms on the config-5 batch, the
  // full row in 64 lanes + the rest alone: see DESIGN.md).  So when the dims do not fill their last group at least half,
  // the streams are packed greedily (widest first) into the full groups and the others run on their own as before.
  StreamMap smap;
  WinSet ws_merged;
  Problem p_merged;
  bool have_merged = false;
  int piece_stream = -1, piece_first = 0;  // stream cut between the merged launch (dims < piece_first) and its own launch
  // Global (D,) and unit variances (round 5): the same packing, on the constant-coefficient kernel (one workgroup walks one
  // (utterance, group of 64 lanes) sequence: 60 + 1 + 3 dims of a Merlin-style row in one group, the other 2 bap dims as a piece).
  const bool merge_strip = (algo == MLPG_HIP_ALGO_AUTO || algo == MLPG_HIP_ALGO_STRIP) && var_mode == MLPG_HIP_VAR_FRAME;
  const bool merge_const = (algo == MLPG_HIP_ALGO_AUTO || algo == MLPG_HIP_ALGO_CONST) &&
                           (var_mode == MLPG_HIP_VAR_GLOBAL || var_mode == MLPG_HIP_VAR_UNIT);
  if (merge_strip || merge_const) {
    int first = -1, cnt = 0, total = 0;
    int members[64];
    size_t coff_first = 0;
    for (int k = 0; k < num_streams; ++k) {
      const mlpg_hip_stream_t &sm = streams_h[k];
      if (sm.static_dim <= 0 || sm.num_windows != 3) continue;
      size_t coff = 0;
      for (int w = 0; w < sm.win_first; ++w) coff += (size_t)(win_l_h[w] + win_u_h[w] + 1);
      bool same = true;
      size_t nco = 0;
      for (int w = 0; w < 3 && same; ++w) {
        const int l = win_l_h[sm.win_first + w], u = win_u_h[sm.win_first + w];
        if (l < 0 || u < 0 || l > 1 || u > 1) same = false;
        else if (first >= 0 && (l != win_l_h[streams_h[first].win_first + w] || u != win_u_h[streams_h[first].win_first + w])) same = false;
        else nco += (size_t)(l + u + 1);
      }
      if (same && first >= 0) same = memcmp(win_coef_h + coff, win_coef_h + coff_first, nco * sizeof(double)) == 0;
      if (!same) continue;
      if (first < 0) { first = k; coff_first = coff; }
      members[cnt++] = k;
      total += sm.static_dim;
    }
    // (Round 6 measured NOT merging for global / unit variances when every narrow member would take the transposed strip form on its own:
    // config 5 in one call 0.73 ms against 0.64-0.65 ms merged -- although the same three launches as three calls sum to 0.61 ms.  Merged.)
    int cap = total;
    if (total > 64 && total % 64 < 32) cap = total - total % 64;
    // widest first (insertion sort, stable), then greedy
    for (int i = 1; i < cnt; ++i)
      for (int j = i; j > 0 && streams_h[members[j]].static_dim > streams_h[members[j - 1]].static_dim; --j) {
        const int t_ = members[j]; members[j] = members[j - 1]; members[j - 1] = t_;
      }
    memset(&smap, 0, sizeof(smap));
    for (int q = 0; q < 4; ++q) smap.begin[q] = INT_MAX;
    int n = 0, pos = 0;
    for (int i = 0; i < cnt && n < 4; ++i) {
      const mlpg_hip_stream_t &sm = streams_h[members[i]];
      if (pos + sm.static_dim > cap) continue;
      smap.begin[n] = pos;
      smap.in_col[n] = sm.in_col;
      smap.sd[n] = sm.static_dim;
      smap.out_col[n] = sm.out_col;
      smap.stat_col[n] = status_cols[members[i]];
      merged_flag[members[i]] = true;
      pos += sm.static_dim;
      ++n;
    }
    // lanes left over: the first dims of one more stream; the rest of it runs as a piece (wave-per-system kernel)
    if (pos < cap && n < 4)
      for (int i = 0; i < cnt; ++i) {
        const int k = members[i];
        if (merged_flag[k]) continue;
        const mlpg_hip_stream_t &sm = streams_h[k];
        smap.begin[n] = pos;
        smap.in_col[n] = sm.in_col;
        smap.sd[n] = sm.static_dim;
        smap.out_col[n] = sm.out_col;
        smap.stat_col[n] = status_cols[k];
        piece_stream = k;
        piece_first = cap - pos;
        pos = cap;
        ++n;
        break;
      }
    smap.n = n;
    smap.total = pos;
    bool ok = n >= 2 && (pos + 63) / 64 <= n &&
              pack_windows(3, win_l_h + streams_h[first].win_first, win_u_h + streams_h[first].win_first, win_coef_h + coff_first, &ws_merged) == 0;
    if (ok) {
      Problem &p = p_merged;
      p.mean = mean;
      p.var = var;
      p.grad_out = nullptr;
      p.lengths = lengths;
      p.out = out;
      p.status = status;
      p.var_mode = var_mode;
      p.B = B;
      p.Tmax = Tmax;
      p.sd = pos;
      p.D = 3 * pos;
      p.ld_in = ld_in;
      p.ld_gout = 0;
      p.ld_out = ld_out;
      p.ld_status = (int)sd_total;
      if (merge_strip) {
        // the decision the widest group would get alone (long utterances, or enough 64-frame strips)
        Problem pw = p;
        pw.sd = pos < 64 ? pos : 64;
        pw.D = 3 * pw.sd;
        ok = strip_supported(p, ws_merged) && (algo == MLPG_HIP_ALGO_STRIP || strip_preferred(pw, ws_merged, false, dtype));
      } else {
        // the constant-coefficient kernel's conditions (const_supported / const_preferred for groups of 64 lanes): a dynamic
        // window of extent 1, and about a sequence per CU
        ok = ws_merged.mw == 1 && rows_fit_buffer(p) && (algo == MLPG_HIP_ALGO_CONST || (long)B * ((pos + 63) / 64) >= 192);
      }
    }
    have_merged = ok;
    if (!ok) {
      memset(merged_flag, 0, sizeof(merged_flag));
      piece_stream = -1;
    }
  }
  int widest = -1;
  for (int k = 0; k < num_streams && !have_merged; ++k)  // (with a merged launch, that one takes the caller's stream)
    if (streams_h[k].static_dim > 0 && !merged_flag[k] &&
        (widest < 0 || streams_h[k].static_dim * (streams_h[k].num_windows + 1) >
                           streams_h[widest].static_dim * (streams_h[widest].num_windows + 1)))
      widest = k;
  hipStream_t main_st = (hipStream_t)stream;
  SideStreams *side = side_streams(device);
  // one caller at a time per device: the side streams and their fork/join events are shared
  std::unique_lock<std::mutex> side_lock;
  if (side) side_lock = std::unique_lock<std::mutex>(side->mu);
  int nside = 0;
  // The fork is recorded BEFORE anything of this call is queued on the caller's stream and the widest stream is
  // launched last: the narrow streams' kernels then only wait for what preceded the call, not for the wide kernel.
  // streams (and the piece) that take the transposed strip form: on the caller's stream, behind the merged launch / in front of the
  // widest stream (see stream_takes_tr)
  bool tr_flag[64] = {};
  for (int k = 0; k < num_streams; ++k) {
    const mlpg_hip_stream_t &sm = streams_h[k];
    if (sm.static_dim <= 0 || merged_flag[k]) continue;
    const bool piece = k == piece_stream;
    tr_flag[k] = stream_takes_tr(dtype, algo, mean, var, var_mode, (long)ld_in, lengths, B, Tmax, sm, win_l_h, win_u_h, win_coef_h, out,
                                 (long)ld_out, piece ? piece_first : 0, piece ? sm.static_dim - piece_first : -1);
  }
  int n_narrow = 0;
  for (int k = 0; k < num_streams; ++k) n_narrow += streams_h[k].static_dim > 0 && k != widest && !merged_flag[k] && !tr_flag[k];
  if (side && n_narrow > 0) MLPG_HIP_CHECK(hipEventRecord(side->fork, main_st));
  auto join_side = [&]() -> int {  // also on the error paths: an unjoined side stream would break a graph capture
    for (int q = 0; q < nside; ++q) {
      MLPG_HIP_CHECK(hipEventRecord(side->join[q], side->st[q]));
      MLPG_HIP_CHECK(hipStreamWaitEvent(main_st, side->join[q], 0));
    }
    nside = 0;
    return 0;
  };
  auto run_stream = [&](const int k, hipStream_t st) -> int {
    const mlpg_hip_stream_t &sm = streams_h[k];
    const bool piece = k == piece_stream;
    if (int rc = stream_entry(device, st, dtype, algo, mean, var, var_mode, (long)ld_in, lengths, B, Tmax, sm, win_l_h,
                              win_u_h, win_coef_h, out, (long)ld_out, status, (int)sd_total, status_cols[k],
                              piece ? piece_first : 0, piece ? sm.static_dim - piece_first : -1))
      return rc;
    if (sm.num_windows == 0 && status) {
      // pass-through streams cannot fail: their status columns are cleared
      MLPG_HIP_CHECK(hipMemset2DAsync(status + status_cols[k], sizeof(int32_t) * (size_t)sd_total, 0,
                                      sizeof(int32_t) * (size_t)sm.static_dim, (size_t)B, st));
    }
    return 0;
  };
  for (int k = 0; k < num_streams; ++k) {
    if (streams_h[k].static_dim <= 0 || k == widest || merged_flag[k] || tr_flag[k]) continue;
    hipStream_t st = main_st;
    if (side && nside < SideStreams::kN) {
      st = side->st[nside];
      if (hipStreamWaitEvent(st, side->fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        st = main_st;
      } else {
        ++nside;
      }
    }
    if (int rc = run_stream(k, st)) {
      (void)join_side();
      return rc;
    }
  }
  if (have_merged) {
    int rc = merge_strip ? launch_strip_multi(main_st, dtype, p_merged, ws_merged, smap, device)
                         : launch_const_multi(main_st, dtype, p_merged, ws_merged, smap, device);
    if (rc == kStripMultiNotResident) {  // the grid cannot hold an utterance (nothing enqueued): one stream after the other
      rc = 0;
      for (int k = 0; k < num_streams && rc == 0; ++k)
        if (merged_flag[k]) rc = run_stream(k, main_st);
      if (rc == 0 && piece_stream >= 0) {  // and the head of the cut stream
        const mlpg_hip_stream_t &sm = streams_h[piece_stream];
        rc = stream_entry(device, main_st, dtype, algo, mean, var, var_mode, (long)ld_in, lengths, B, Tmax, sm, win_l_h, win_u_h,
                          win_coef_h, out, (long)ld_out, status, (int)sd_total, status_cols[piece_stream], 0, piece_first);
      }
    }
    if (rc) {
      (void)join_side();
      return rc;
    }
  }
  for (int k = 0; k < num_streams; ++k)
    if (tr_flag[k] && k != widest)
      if (int rc = run_stream(k, main_st)) {
        (void)join_side();
        return rc;
      }
  if (widest >= 0)
    if (int rc = run_stream(widest, main_st)) {
      (void)join_side();
      return rc;
    }
  if (int rc = join_side()) return rc;
  return 0;
}

namespace {
// The form mlpg_hip_unit_mse_step takes for a problem -- decided HERE and nowhere else (the step and mlpg_hip_unit_mse_form both ask):
// 2 the FIR form (two launches; needs the workspace of mlpg_hip_unit_mse_workspace_bytes_t), 1 the one-launch wave-per-system kernel,
// 0 neither.  `with_fir_workspace`: the caller can provide the larger workspace.
int unit_mse_form(hipStream_t st, int device, int dtype, const Problem &p, const WinSet &ws, bool with_fir_workspace) {
  if (with_fir_workspace && dtype == MLPG_HIP_F32 && fir_shape_supported(p, ws, dtype, dtype) && fir_table_ready(st, device, ws)) return 2;
  return unit_mse_supported(p.Tmax, ws) ? 1 : 0;
}
}  // namespace

__attribute__((visibility("default"))) int mlpg_hip_unit_mse_form(int device, void *stream, int dtype, int has_lengths, int B,
                                                                  int Tmax, int D, int num_windows, const int32_t *win_l_h,
                                                                  const int32_t *win_u_h, const double *win_coef_h) {
  if (int rc = check_common(B, Tmax, D, num_windows)) return rc;
  if (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  WinSet ws;
  if (int rc = pack_windows(num_windows, win_l_h, win_u_h, win_coef_h, &ws)) return rc;
  if (B == 0 || Tmax == 0 || D == 0) return 1;  // (the step answers an empty batch itself: loss 0)
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  Problem p = {};
  static const int32_t some_lengths = 0;
  p.lengths = has_lengths ? &some_lengths : nullptr;  // (only looked at as "given or not")
  p.var_mode = MLPG_HIP_VAR_UNIT;
  p.B = B;
  p.Tmax = Tmax;
  p.D = D;
  p.sd = D / num_windows;
  p.ld_in = D;
  p.ld_out = D;
  p.ld_status = D / num_windows;
  return unit_mse_form((hipStream_t)stream, device, dtype, p, ws, true);
}

__attribute__((visibility("default"))) int mlpg_hip_unit_mse_step(int device, void *stream, int dtype, const void *mean,
                                                                  const void *target, const int32_t *lengths, int B,
                                                                  int Tmax, int D, int num_windows,
                                                                  const int32_t *win_l_h, const int32_t *win_u_h,
                                                                  const double *win_coef_h, double n_elems, void *y_out,
                                                                  void *grad_mean, double *loss, int32_t *status,
                                                                  void *workspace, size_t workspace_bytes) {
  if (int rc = check_common(B, Tmax, D, num_windows)) return rc;
  if (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  if (!(n_elems > 0.0)) {
    set_error("unit_mse_step: n_elems must be positive");
    return MLPG_HIP_EINVAL;
  }
  WinSet ws;
  if (int rc = pack_windows(num_windows, win_l_h, win_u_h, win_coef_h, &ws)) return rc;
  if (!loss) {
    set_error("unit_mse_step: NULL loss pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || Tmax == 0 || D == 0) {
    MLPG_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(double), st));
    return 0;
  }
  if (!mean || !target || !grad_mean) {
    set_error("NULL data pointer");
    return MLPG_HIP_EINVAL;
  }
  Problem p;
  p.mean = mean;
  p.var = nullptr;
  p.grad_out = nullptr;
  p.lengths = lengths;
  p.out = grad_mean;
  p.status = status;
  p.var_mode = MLPG_HIP_VAR_UNIT;
  p.B = B;
  p.Tmax = Tmax;
  p.D = D;
  p.sd = D / num_windows;
  p.ld_in = D;
  p.ld_gout = 0;
  p.ld_out = D;
  p.ld_status = D / num_windows;
  // float32 batches without lengths, given the larger workspace (mlpg_hip_unit_mse_workspace_bytes_t): the FIR form, two launches
  const bool fir_ws = workspace && !((uintptr_t)workspace & 127) && workspace_bytes >= fir_mse_workspace_bytes(B, Tmax, p.sd);
  const int form = unit_mse_form(st, device, dtype, p, ws, fir_ws);
  if (form == 2) {
    const int rc = launch_fir_mse(st, p, ws, device, target, y_out, n_elems, loss, workspace);
    if (rc != kFirNotApplicable) return rc;
  }
  if (!unit_mse_supported(Tmax, ws)) {
    set_error("unit_mse_step: needs window extents <= 1 and T <= 1024 (T=%d, half-bandwidth %d), or a float32 batch without lengths "
              "of T >= 96 whose window set passes the FIR form's decay test (mlpg_hip_unit_mse_form says which), with the workspace of "
              "mlpg_hip_unit_mse_workspace_bytes_t; inside a stream capture the FIR form needs its tap table built by an earlier call",
              Tmax, ws.q);
    return MLPG_HIP_EINVAL;
  }
  if (!workspace || workspace_bytes < unit_mse_workspace_bytes(B, D / num_windows) || ((uintptr_t)workspace & 127)) {
    set_error("unit_mse_step: workspace of %zu bytes (128-byte aligned) needed, see mlpg_hip_unit_mse_workspace_bytes",
              unit_mse_workspace_bytes(B, D / num_windows));
    return MLPG_HIP_EINVAL;
  }
  return launch_unit_mse(st, dtype, p, ws, target, y_out, n_elems, loss, workspace);
}

__attribute__((visibility("default"))) size_t mlpg_hip_unit_mse_workspace_bytes(int B, int D, int num_windows) {
  if (B < 0 || D < 0 || num_windows < 1) return 0;
  return unit_mse_workspace_bytes(B, D / num_windows);
}

__attribute__((visibility("default"))) size_t mlpg_hip_unit_mse_workspace_bytes_t(int B, int Tmax, int D, int num_windows) {
  if (B < 0 || Tmax < 0 || D < 0 || num_windows < 1) return 0;
  return std::max(unit_mse_workspace_bytes(B, D / num_windows), fir_mse_workspace_bytes(B, Tmax, D / num_windows));
}

__attribute__((visibility("default"))) int mlpg_hip_stream_copy(int device, void *stream, const void *src, void *dst,
                                                                size_t nbytes) {
  if (nbytes == 0) return 0;
  if (!src || !dst || (((uintptr_t)src | (uintptr_t)dst) & 15)) {
    set_error("stream_copy: NULL or not 16-byte aligned pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_stream_copy((hipStream_t)stream, src, dst, nbytes);
}

__attribute__((visibility("default"))) int mlpg_hip_backward(int device, void *stream, int in_dtype, int out_dtype,
                                                             int algo, const void *var, int var_mode,
                                                             const void *grad_out, const int32_t *lengths, int B,
                                                             int Tmax, int D, int num_windows,
                                                             const int32_t *win_l_h, const int32_t *win_u_h,
                                                             const double *win_coef_h, void *grad_mean,
                                                             int32_t *status) {
  return solve_entry(device, stream, in_dtype, out_dtype, algo, true, nullptr, var, var_mode, grad_out, lengths, B,
                     Tmax, D, num_windows, win_l_h, win_u_h, win_coef_h, grad_mean, status);
}

__attribute__((visibility("default"))) int mlpg_hip_delta_features(int device, void *stream, int dtype, const void *x,
                                                                   const int32_t *lengths, int B, int Tmax, int D,
                                                                   int num_windows, const int32_t *win_l_h,
                                                                   const int32_t *win_u_h, const double *win_coef_h,
                                                                   void *out) {
  if (B < 0 || Tmax < 0 || D < 0 || (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64)) {
    set_error("delta_features: bad arguments");
    return MLPG_HIP_EINVAL;
  }
  WinSet ws;
  if (int rc = pack_windows(num_windows, win_l_h, win_u_h, win_coef_h, &ws)) return rc;
  if ((long)B * Tmax * D == 0) return 0;
  if (!x || !out) {
    set_error("delta_features: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_delta((hipStream_t)stream, dtype, x, lengths, B, Tmax, D, ws, out);
}

namespace {
std::atomic<bool> g_modspec_direct{false};
int modspec_entry(int device, void *stream, int mode, const double *x, const double *ms, const double *ph, double *out,
                  double *out_ph, int B, int T, int D, int n, int ortho, int limit_bin, int log_domain) {
  if (B < 0 || T < 0 || D < 0 || n < 1) {
    set_error("modspec: negative size");
    return MLPG_HIP_EINVAL;
  }
  if (n < 2) {
    set_error("modspec: the DFT length must be at least 2 (got %d)", n);
    return MLPG_HIP_EINVAL;
  }
  if (T > n) {
    set_error("modspec: DFT length %d must not be smaller than the time length %d", n, T);
    return MLPG_HIP_EINVAL;
  }
  if ((long)B * D == 0) return 0;
  if (!out || (mode == 1 ? (!ms || !ph) : !x) || (mode == 3 && !ms)) {
    set_error("modspec: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  // powers of two up to 4096 (the reference's defaults are 2048 and 4096): the fused in-LDS FFT; any other length:
  // the direct transform of modspec_dft.hip
  if (n <= 4096 && !(n & (n - 1)) && !g_modspec_direct)
    return launch_modspec((hipStream_t)stream, mode, x, ms, ph, out, out_ph, B, T, D, n, ortho, limit_bin, log_domain);
  return launch_modspec_dft((hipStream_t)stream, device, mode, x, ms, ph, out, out_ph, B, T, D, n, ortho, limit_bin,
                            log_domain);
}
}  // namespace

// Testing aid: route every DFT length through the direct transform (so that it can be compared with the FFT path).
__attribute__((visibility("default"))) void mlpg_hip_modspec_set_direct(int on) { g_modspec_direct = on != 0; }

__attribute__((visibility("default"))) int mlpg_hip_modspec(int device, void *stream, const double *x, int B, int T,
                                                            int D, int n, int ortho, double *ms, double *phase) {
  return modspec_entry(device, stream, 0, x, nullptr, nullptr, ms, phase, B, T, D, n, ortho, 0, 0);
}

__attribute__((visibility("default"))) int mlpg_hip_inv_modspec(int device, void *stream, const double *ms,
                                                                const double *phase, int B, int n, int D, int ortho,
                                                                double *x) {
  return modspec_entry(device, stream, 1, nullptr, ms, phase, x, nullptr, B, 0, D, n, ortho, 0, 0);
}

__attribute__((visibility("default"))) int mlpg_hip_modspec_smoothing(int device, void *stream, const double *x, int B,
                                                                      int T, int D, int n, int ortho, int limit_bin,
                                                                      int log_domain, double *out) {
  return modspec_entry(device, stream, 2, x, nullptr, nullptr, out, nullptr, B, T, D, n, ortho, limit_bin, log_domain);
}

__attribute__((visibility("default"))) int mlpg_hip_modspec_backward(int device, void *stream, const double *x,
                                                                     const double *grad_ms, int B, int T, int D,
                                                                     int n, int ortho, double *grad_x) {
  return modspec_entry(device, stream, 3, x, grad_ms, nullptr, grad_x, nullptr, B, T, D, n, ortho, 0, 0);
}

__attribute__((visibility("default"))) int mlpg_hip_trim_lengths(int device, void *stream, int dtype, const void *X,
                                                                 int N, int T, int D, double eps,
                                                                 int32_t *lengths) {
  if (N < 0 || T < 0 || D < 0 || (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64)) {
    set_error("trim_lengths: bad arguments");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  if (!lengths || (!X && (long)T * D > 0)) {
    set_error("trim_lengths: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_trim((hipStream_t)stream, dtype, X, N, T, D, eps, lengths);
}

__attribute__((visibility("default"))) int mlpg_hip_fastdtw(int device, void *stream, const double *X,
                                                            const double *Y, const int32_t *lenx,
                                                            const int32_t *leny, int N, int Tx, int Ty, int D,
                                                            int radius, int dist_kind, double dist_scale,
                                                            int tie_rule, int32_t *path_i, int32_t *path_j,
                                                            int32_t *path_len, double *cost) {
  if (N < 0 || Tx < 1 || Ty < 1 || D < 1 || radius < 1) {
    set_error("fastdtw: need N >= 0, Tx, Ty, D >= 1 and radius >= 1");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  if (!X || !Y || !lenx || !leny || !path_i || !path_j || !path_len || !cost) {
    set_error("fastdtw: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_fastdtw((hipStream_t)stream, device, X, Y, lenx, leny, N, Tx, Ty, D, radius, dist_kind, dist_scale,
                        tie_rule, path_i, path_j, path_len, cost);
}

__attribute__((visibility("default"))) int mlpg_hip_dtw_level_windows(int device, void *stream, int N, int radius,
                                                                      const int32_t *level_tx, const int32_t *level_ty,
                                                                      const int32_t *full, const int32_t *cpath_i,
                                                                      const int32_t *cpath_j, const int32_t *cpath_len,
                                                                      int cpath_stride, int32_t *row_lo, int32_t *row_hi,
                                                                      int64_t *row_off, int row_stride) {
  if (N < 0 || radius < 1 || row_stride < 1 || (N > 0 && (!level_tx || !level_ty || !full || !row_lo || !row_hi || !row_off))) {
    set_error("dtw_level_windows: bad argument");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_dtw_window((hipStream_t)stream, device, N, radius, level_tx, level_ty, full, cpath_i, cpath_j, cpath_len,
                           cpath_stride, row_lo, row_hi, row_off, row_stride);
}

__attribute__((visibility("default"))) int mlpg_hip_dtw_level_from_costs(int device, void *stream, int N, int tie_rule,
                                                                         const int32_t *level_tx, const int32_t *level_ty,
                                                                         const int32_t *row_lo, const int32_t *row_hi,
                                                                         const int64_t *row_off, int row_stride, int max_ty,
                                                                         const double *costs, const int64_t *cost_base,
                                                                         int64_t total_cells, int32_t *path_i,
                                                                         int32_t *path_j, int32_t *path_len,
                                                                         int path_stride, double *cost) {
  if (N < 0 || row_stride < 1 || max_ty < 1 || path_stride < 1 || total_cells < 0 ||
      (tie_rule != MLPG_HIP_TIE_FIRST_MIN && tie_rule != MLPG_HIP_TIE_DIAG_LAST) ||
      (N > 0 && (!level_tx || !level_ty || !row_lo || !row_hi || !row_off || !costs || !cost_base || !path_i || !path_j ||
                 !path_len || !cost))) {
    set_error("dtw_level_from_costs: bad argument");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_dtw_costs((hipStream_t)stream, device, N, tie_rule, level_tx, level_ty, row_lo, row_hi, row_off, row_stride,
                          max_ty, costs, cost_base, total_cells, path_i, path_j, path_len, path_stride, cost);
}

__attribute__((visibility("default"))) int mlpg_hip_fastdtw_l2(int device, void *stream, const double *X,
                                                               const double *Y, const int32_t *lenx,
                                                               const int32_t *leny, int N, int Tx, int Ty, int D,
                                                               int radius, int32_t *path_i, int32_t *path_j,
                                                               int32_t *path_len, double *cost) {
  return mlpg_hip_fastdtw(device, stream, X, Y, lenx, leny, N, Tx, Ty, D, radius, MLPG_HIP_DIST_L2, 1.0,
                          MLPG_HIP_TIE_FIRST_MIN, path_i, path_j, path_len, cost);
}

__attribute__((visibility("default"))) int mlpg_hip_gmm_convert(int device, void *stream, const double *x,
                                                                const double *posterior, const int32_t *mixture,
                                                                const double *mu_x, const double *mu_y,
                                                                const double *A, int64_t N, int D, int Dy, int M,
                                                                double *out) {
  if (N < 0 || D < 1 || Dy < 1 || M < 1) {
    set_error("gmm_convert: bad sizes");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  if (!x || !mu_x || !mu_y || !A || !out || (!posterior && !mixture)) {
    set_error("gmm_convert: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_gmm_convert((hipStream_t)stream, x, posterior, mixture, mu_x, mu_y, A, (long)N, D, Dy, M, out);
}

__attribute__((visibility("default"))) int mlpg_hip_gather_path(int device, void *stream, int dtype, const void *src,
                                                                const int32_t *path, const int32_t *path_len, int N,
                                                                int Tsrc, int path_stride, int D, int Tout,
                                                                void *out) {
  if (N < 0 || Tsrc < 0 || D < 0 || Tout < 0 || path_stride < 0 ||
      (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64)) {
    set_error("gather_path: bad arguments");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0 || Tout == 0 || D == 0) return 0;
  if (!src || !path || !path_len || !out) {
    set_error("gather_path: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  DeviceGuard g(device);
  if (!g.ok) {
    set_error("cannot select device %d", device);
    return MLPG_HIP_ERUNTIME;
  }
  return launch_gather((hipStream_t)stream, dtype, src, path, path_len, N, Tsrc, path_stride, D, Tout, out);
}

}  // extern "C"
