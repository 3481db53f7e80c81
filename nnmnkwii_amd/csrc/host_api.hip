// Host-pointer entry points: numpy-in / numpy-out calls without any framework tensor in between.
//
// mlpg_hip_forward_host replaces the loop a user of the reference writes around paramgen.mlpg for a padded batch
// held in HOST memory (util/__init__.py:44-66 over datasets/__init__.py:152-218 arrays).  The batch is cut into
// chunks of utterances that alternate between two HIP streams; per chunk: host -> device copy, the MLPG kernels,
// device -> host copy, all asynchronous, so that the PCIe transfers of one chunk run under the kernels of the other
// and under the CPU-side staging of the next.  Pageable host memory is staged through pinned buffers by a few copy
// threads (a single-threaded memcpy is slower than PCIe Gen5); memory that is already pinned (mlpg_hip_host_alloc,
// hipHostMalloc, hipHostRegister) is transferred in place.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace mlpg {

int dispatch_solve(hipStream_t st, int in_dtype, int out_dtype, int algo, bool backward, const Problem &p,
                   const WinSet &ws, int device);
int pack_windows_public(int nw, const int32_t *wl, const int32_t *wu, const double *wc, WinSet *ws);

namespace {

struct HostCtx {
  hipStream_t st[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  void *pin_in[2] = {nullptr, nullptr};   // staged mean | var of one chunk
  void *pin_out[2] = {nullptr, nullptr};  // staged out | status of one chunk
  size_t pin_in_bytes = 0, pin_out_bytes = 0;
  void *dev[2] = {nullptr, nullptr};      // mean | var | out | status | lengths of one chunk
  size_t dev_bytes = 0;
  bool ok = false;
};
HostCtx g_host[16];
std::mutex g_host_mu;

bool is_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

void parallel_copy(void *dst, const void *src, size_t bytes) {
  const size_t kMin = 4u << 20;
  unsigned nt = std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
  if (bytes < 2 * kMin || nt == 1) {
    memcpy(dst, src, bytes);
    return;
  }
  nt = (unsigned)std::min<size_t>(nt, bytes / kMin);
  std::vector<std::thread> th;
  const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
  for (unsigned t = 1; t < nt; ++t) {
    const size_t off = per * t;
    if (off >= bytes) break;
    const size_t n = std::min(per, bytes - off);
    th.emplace_back([=] { memcpy((char *)dst + off, (const char *)src + off, n); });
  }
  memcpy(dst, src, std::min(per, bytes));
  for (auto &t : th) t.join();
}

int ensure(HostCtx &c, size_t in_bytes, size_t out_bytes, size_t dev_bytes) {
  if (!c.ok) {
    for (int k = 0; k < 2; ++k) {
      MLPG_HIP_CHECK(hipStreamCreateWithFlags(&c.st[k], hipStreamNonBlocking));
      MLPG_HIP_CHECK(hipEventCreateWithFlags(&c.done[k], hipEventDisableTiming));
    }
    c.ok = true;
  }
  auto grow_pin = [&](void *(&buf)[2], size_t &have, size_t want) -> int {
    if (have >= want) return 0;
    for (int k = 0; k < 2; ++k) {
      if (buf[k]) (void)hipHostFree(buf[k]);
      buf[k] = nullptr;
      MLPG_HIP_CHECK(hipHostMalloc(&buf[k], want, hipHostMallocDefault));
    }
    have = want;
    return 0;
  };
  if (int rc = grow_pin(c.pin_in, c.pin_in_bytes, in_bytes)) return rc;
  if (int rc = grow_pin(c.pin_out, c.pin_out_bytes, out_bytes)) return rc;
  if (c.dev_bytes < dev_bytes) {
    for (int k = 0; k < 2; ++k) {
      if (c.dev[k]) (void)hipFree(c.dev[k]);
      c.dev[k] = nullptr;
      MLPG_HIP_CHECK(hipMalloc(&c.dev[k], dev_bytes));
    }
    c.dev_bytes = dev_bytes;
  }
  return 0;
}

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

__global__ void widen_f32(const float *__restrict__ src, double *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

}  // namespace

void host_api_shutdown() {
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (int d = 0; d < 16; ++d) {
    HostCtx &c = g_host[d];
    if (!c.ok && !c.dev[0] && !c.pin_in[0] && !c.pin_out[0]) continue;
    (void)hipSetDevice(d);
    for (int k = 0; k < 2; ++k) {
      if (c.st[k]) {
        (void)hipStreamSynchronize(c.st[k]);
        (void)hipStreamDestroy(c.st[k]);
      }
      if (c.done[k]) (void)hipEventDestroy(c.done[k]);
      if (c.pin_in[k]) (void)hipHostFree(c.pin_in[k]);
      if (c.pin_out[k]) (void)hipHostFree(c.pin_out[k]);
      if (c.dev[k]) (void)hipFree(c.dev[k]);
    }
    c = HostCtx();
  }
}
}  // namespace mlpg

using namespace mlpg;

extern "C" {

__attribute__((visibility("default"))) void *mlpg_hip_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    set_error("host_alloc: hipHostMalloc of %zu bytes failed", bytes);
    return nullptr;
  }
  return p;
}

__attribute__((visibility("default"))) void mlpg_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

__attribute__((visibility("default"))) int mlpg_hip_forward_host(int device, int dtype, int algo, const void *mean_h,
                                                                 const void *var_h, int var_mode,
                                                                 const int32_t *lengths_h, int B, int Tmax, int D,
                                                                 int num_windows, const int32_t *win_l_h,
                                                                 const int32_t *win_u_h, const double *win_coef_h,
                                                                 void *out_h, int32_t *status_h) {
  if (B < 0 || Tmax < 0 || D < 0 || num_windows < 1 || D % num_windows != 0) {
    set_error("forward_host: bad sizes (B=%d, Tmax=%d, D=%d, num_windows=%d)", B, Tmax, D, num_windows);
    return MLPG_HIP_EINVAL;
  }
  if (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  if (var_mode < 0 || var_mode > 2 || (var_mode != MLPG_HIP_VAR_UNIT && !var_h)) {
    set_error("bad var_mode %d / NULL var", var_mode);
    return MLPG_HIP_EINVAL;
  }
  WinSet ws;
  if (int rc = pack_windows_public(num_windows, win_l_h, win_u_h, win_coef_h, &ws)) return rc;
  if ((long)B * Tmax * D == 0) return 0;
  if (!mean_h || !out_h) {
    set_error("NULL data pointer");
    return MLPG_HIP_EINVAL;
  }
  if (device < 0 || device >= 16) {
    set_error("bad device %d", device);
    return MLPG_HIP_EINVAL;
  }
  int prev = -1;
  MLPG_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) MLPG_HIP_CHECK(hipSetDevice(device));
  struct Restore {
    int prev, dev;
    ~Restore() { if (prev != dev) (void)hipSetDevice(prev); }
  } restore{prev, device};

  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8;
  const int sd = D / num_windows;
  const size_t utt_in = (size_t)Tmax * D * esz, utt_out = (size_t)Tmax * sd * esz;
  const bool fvar = var_mode == MLPG_HIP_VAR_FRAME;
  // ~64 MB of input per chunk, at least 4 chunks when the batch allows (so that transfers and kernels overlap)
  static const long chunk_mb = [] { const char *e = getenv("MLPG_HIP_HOST_CHUNK_MB"); const long v = e ? atol(e) : 0; return v > 0 ? v : 64; }();
  long cb = (long)(((size_t)chunk_mb << 20) / (utt_in * (fvar ? 2 : 1)));
  cb = std::max<long>(1, std::min<long>(cb, (B + 3) / 4));
  const size_t in_bytes = (size_t)cb * utt_in * (fvar ? 2 : 1);
  const size_t out_bytes = up256((size_t)cb * utt_out) + (size_t)cb * sd * sizeof(int32_t);
  const size_t o_mean = 0, o_var = up256((size_t)cb * utt_in), o_out = o_var + (fvar ? up256((size_t)cb * utt_in) : up256((size_t)D * esz)),
               o_status = o_out + up256((size_t)cb * utt_out), o_len = o_status + up256((size_t)cb * sd * sizeof(int32_t)),
               dev_bytes = o_len + up256((size_t)cb * sizeof(int32_t));

  std::lock_guard<std::mutex> lk(g_host_mu);  // one host call at a time per process (the staging buffers are shared)
  HostCtx &c = g_host[device];
  if (int rc = ensure(c, in_bytes, out_bytes, dev_bytes)) return rc;
  const bool mean_pinned = is_pinned(mean_h), var_pinned = fvar && is_pinned(var_h), out_pinned = is_pinned(out_h);

  struct Pending {
    bool active = false;
    long b0 = 0, nb = 0;
  } pend[2];
  auto finish = [&](int slot) -> int {  // wait for the slot's chunk, hand its staged results to the caller
    if (!pend[slot].active) return 0;
    MLPG_HIP_CHECK(hipEventSynchronize(c.done[slot]));
    const long b0 = pend[slot].b0, nb = pend[slot].nb;
    if (!out_pinned) parallel_copy((char *)out_h + (size_t)b0 * utt_out, c.pin_out[slot], (size_t)nb * utt_out);
    if (status_h) memcpy(status_h + (size_t)b0 * sd, (char *)c.pin_out[slot] + up256((size_t)cb * utt_out), (size_t)nb * sd * sizeof(int32_t));
    pend[slot].active = false;
    return 0;
  };

  if (var_mode == MLPG_HIP_VAR_GLOBAL) {
    for (int k = 0; k < 2; ++k)
      MLPG_HIP_CHECK(hipMemcpyAsync((char *)c.dev[k] + o_var, var_h, (size_t)D * esz, hipMemcpyHostToDevice, c.st[k]));
  }
  int chunk = 0;
  for (long b0 = 0; b0 < B; b0 += cb, ++chunk) {
    const int slot = chunk & 1;
    const long nb = std::min<long>(cb, B - b0);
    if (int rc = finish(slot)) return rc;  // the slot's buffers are free again
    hipStream_t st = c.st[slot];
    char *d = (char *)c.dev[slot];
    const char *msrc = (const char *)mean_h + (size_t)b0 * utt_in;
    if (!mean_pinned) {
      parallel_copy(c.pin_in[slot], msrc, (size_t)nb * utt_in);
      msrc = (const char *)c.pin_in[slot];
    }
    MLPG_HIP_CHECK(hipMemcpyAsync(d + o_mean, msrc, (size_t)nb * utt_in, hipMemcpyHostToDevice, st));
    if (fvar) {
      const char *vsrc = (const char *)var_h + (size_t)b0 * utt_in;
      if (!var_pinned) {
        char *stage = (char *)c.pin_in[slot] + (size_t)cb * utt_in;
        parallel_copy(stage, vsrc, (size_t)nb * utt_in);
        vsrc = stage;
      }
      MLPG_HIP_CHECK(hipMemcpyAsync(d + o_var, vsrc, (size_t)nb * utt_in, hipMemcpyHostToDevice, st));
    }
    if (lengths_h) MLPG_HIP_CHECK(hipMemcpyAsync(d + o_len, lengths_h + b0, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice, st));
    Problem p;
    p.mean = d + o_mean;
    p.var = var_mode == MLPG_HIP_VAR_UNIT ? nullptr : d + o_var;
    p.grad_out = nullptr;
    p.lengths = lengths_h ? (const int32_t *)(d + o_len) : nullptr;
    p.out = d + o_out;
    p.status = (int32_t *)(d + o_status);
    p.var_mode = var_mode;
    p.B = (int)nb;
    p.Tmax = Tmax;
    p.D = D;
    p.sd = sd;
    p.ld_in = D;
    p.ld_gout = 0;
    p.ld_out = sd;
    p.ld_status = sd;
    if (int rc = dispatch_solve(st, dtype, dtype, algo, false, p, ws, device)) return rc;
    void *odst = out_pinned ? (void *)((char *)out_h + (size_t)b0 * utt_out) : c.pin_out[slot];
    MLPG_HIP_CHECK(hipMemcpyAsync(odst, d + o_out, (size_t)nb * utt_out, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync((char *)c.pin_out[slot] + up256((size_t)cb * utt_out), d + o_status,
                                  (size_t)nb * sd * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipEventRecord(c.done[slot], st));
    pend[slot].active = true;
    pend[slot].b0 = b0;
    pend[slot].nb = nb;
  }
  if (int rc = finish(chunk & 1)) return rc;
  if (int rc = finish((chunk + 1) & 1)) return rc;
  return 0;
}

// fastdtw for N utterance pairs held in HOST memory: what DTWAligner.transform (preprocessing/alignment.py:41-76) does
// per pair -- trim_zeros_frames on both utterances (:46-47, when lenx_h / leny_h are NULL: trailing frames with
// sum |x| < eps are dropped on the device), fastdtw (:50) -- chunked over the pairs like mlpg_hip_forward_host, the
// transfers of one chunk under the kernels of the other.  X_h (N, Tx, D), Y_h (N, Ty, D) float32 or float64 (float32
// is widened on the device: the distances are computed in float64 either way).  Outputs as mlpg_hip_fastdtw;
// lenx_out_h / leny_out_h (may be NULL) receive the lengths that were used.
__attribute__((visibility("default"))) int mlpg_hip_fastdtw_host(int device, int dtype, const void *X_h, const void *Y_h,
                                                                 const int32_t *lenx_h, const int32_t *leny_h, int N,
                                                                 int Tx, int Ty, int D, int radius, int dist_kind,
                                                                 double dist_scale, int tie_rule, double trim_eps,
                                                                 int32_t *path_i_h, int32_t *path_j_h,
                                                                 int32_t *path_len_h, double *cost_h,
                                                                 int32_t *lenx_out_h, int32_t *leny_out_h) {
  if (N < 0 || Tx < 1 || Ty < 1 || D < 1 || radius < 1) {
    set_error("fastdtw_host: need N >= 0, Tx, Ty, D >= 1 and radius >= 1");
    return MLPG_HIP_EINVAL;
  }
  if (dtype != MLPG_HIP_F32 && dtype != MLPG_HIP_F64) {
    set_error("dtype must be MLPG_HIP_F32 or MLPG_HIP_F64");
    return MLPG_HIP_EINVAL;
  }
  if ((lenx_h == nullptr) != (leny_h == nullptr)) {
    set_error("fastdtw_host: give both length arrays or neither");
    return MLPG_HIP_EINVAL;
  }
  if (N == 0) return 0;
  if (!X_h || !Y_h || !path_i_h || !path_j_h || !path_len_h || !cost_h) {
    set_error("fastdtw_host: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  if (device < 0 || device >= 16) {
    set_error("bad device %d", device);
    return MLPG_HIP_EINVAL;
  }
  int prev = -1;
  MLPG_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) MLPG_HIP_CHECK(hipSetDevice(device));
  struct Restore {
    int prev, dev;
    ~Restore() { if (prev != dev) (void)hipSetDevice(prev); }
  } restore{prev, device};

  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8;
  const size_t px = (size_t)Tx * D, py = (size_t)Ty * D, pl = (size_t)Tx + Ty;  // elements per pair: X, Y, path slots
  long cb = (long)((64u << 20) / ((px + py) * esz));
  cb = std::max<long>(1, std::min<long>(cb, (N + 3) / 4));
  // pinned staging: in = X | Y;  out = path_i | path_j | path_len | lenx | leny | cost
  const size_t in_bytes = up256((size_t)cb * px * esz) + (size_t)cb * py * esz;
  const size_t so_pj = up256((size_t)cb * pl * 4), so_pl = 2 * so_pj, so_lx = so_pl + up256((size_t)cb * 4),
               so_ly = so_lx + up256((size_t)cb * 4), so_c = so_ly + up256((size_t)cb * 4),
               out_bytes = so_c + (size_t)cb * 8;
  // device: X | Y | X64 | Y64 (float32 input only) | lenx | leny | path_i | path_j | path_len | cost
  const size_t o_x = 0, o_y = up256((size_t)cb * px * esz), o_x64 = o_y + up256((size_t)cb * py * esz),
               o_y64 = o_x64 + (dtype == MLPG_HIP_F32 ? up256((size_t)cb * px * 8) : 0),
               o_lx = o_y64 + (dtype == MLPG_HIP_F32 ? up256((size_t)cb * py * 8) : 0), o_ly = o_lx + up256((size_t)cb * 4),
               o_pi = o_ly + up256((size_t)cb * 4), o_pj = o_pi + so_pj, o_pl = o_pj + so_pj,
               o_c = o_pl + up256((size_t)cb * 4), dev_bytes = o_c + up256((size_t)cb * 8);

  std::lock_guard<std::mutex> lk(g_host_mu);
  HostCtx &c = g_host[device];
  if (int rc = ensure(c, in_bytes, out_bytes, dev_bytes)) return rc;
  const bool x_pinned = is_pinned(X_h), y_pinned = is_pinned(Y_h);

  struct Pending {
    bool active = false;
    long b0 = 0, nb = 0;
  } pend[2];
  auto finish = [&](int slot) -> int {
    if (!pend[slot].active) return 0;
    MLPG_HIP_CHECK(hipEventSynchronize(c.done[slot]));
    const long b0 = pend[slot].b0, nb = pend[slot].nb;
    const char *o = (const char *)c.pin_out[slot];
    memcpy(path_i_h + (size_t)b0 * pl, o, (size_t)nb * pl * 4);
    memcpy(path_j_h + (size_t)b0 * pl, o + so_pj, (size_t)nb * pl * 4);
    memcpy(path_len_h + b0, o + so_pl, (size_t)nb * 4);
    if (lenx_out_h) memcpy(lenx_out_h + b0, o + so_lx, (size_t)nb * 4);
    if (leny_out_h) memcpy(leny_out_h + b0, o + so_ly, (size_t)nb * 4);
    memcpy(cost_h + b0, o + so_c, (size_t)nb * 8);
    pend[slot].active = false;
    return 0;
  };

  int chunk = 0;
  for (long b0 = 0; b0 < N; b0 += cb, ++chunk) {
    const int slot = chunk & 1;
    const long nb = std::min<long>(cb, N - b0);
    if (int rc = finish(slot)) return rc;
    hipStream_t st = c.st[slot];
    char *d = (char *)c.dev[slot];
    const char *xs = (const char *)X_h + (size_t)b0 * px * esz, *ys = (const char *)Y_h + (size_t)b0 * py * esz;
    if (!x_pinned) {
      parallel_copy(c.pin_in[slot], xs, (size_t)nb * px * esz);
      xs = (const char *)c.pin_in[slot];
    }
    MLPG_HIP_CHECK(hipMemcpyAsync(d + o_x, xs, (size_t)nb * px * esz, hipMemcpyHostToDevice, st));
    if (!y_pinned) {
      char *stage = (char *)c.pin_in[slot] + up256((size_t)cb * px * esz);
      parallel_copy(stage, ys, (size_t)nb * py * esz);
      ys = stage;
    }
    MLPG_HIP_CHECK(hipMemcpyAsync(d + o_y, ys, (size_t)nb * py * esz, hipMemcpyHostToDevice, st));
    int32_t *lx = (int32_t *)(d + o_lx), *ly = (int32_t *)(d + o_ly);
    if (lenx_h) {
      MLPG_HIP_CHECK(hipMemcpyAsync(lx, lenx_h + b0, (size_t)nb * 4, hipMemcpyHostToDevice, st));
      MLPG_HIP_CHECK(hipMemcpyAsync(ly, leny_h + b0, (size_t)nb * 4, hipMemcpyHostToDevice, st));
    } else {
      if (int rc = launch_trim(st, dtype, d + o_x, (int)nb, Tx, D, trim_eps, lx)) return rc;
      if (int rc = launch_trim(st, dtype, d + o_y, (int)nb, Ty, D, trim_eps, ly)) return rc;
    }
    const double *x64 = (const double *)(d + o_x), *y64 = (const double *)(d + o_y);
    if (dtype == MLPG_HIP_F32) {
      hipLaunchKernelGGL(widen_f32, dim3(1024), dim3(256), 0, st, (const float *)(d + o_x), (double *)(d + o_x64), (size_t)nb * px);
      hipLaunchKernelGGL(widen_f32, dim3(1024), dim3(256), 0, st, (const float *)(d + o_y), (double *)(d + o_y64), (size_t)nb * py);
      MLPG_HIP_CHECK(hipGetLastError());
      x64 = (const double *)(d + o_x64);
      y64 = (const double *)(d + o_y64);
    }
    if (int rc = launch_fastdtw(st, device, x64, y64, lx, ly, (int)nb, Tx, Ty, D, radius, dist_kind, dist_scale, tie_rule,
                                (int32_t *)(d + o_pi), (int32_t *)(d + o_pj), (int32_t *)(d + o_pl), (double *)(d + o_c)))
      return rc;
    char *o = (char *)c.pin_out[slot];
    MLPG_HIP_CHECK(hipMemcpyAsync(o, d + o_pi, (size_t)nb * pl * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_pj, d + o_pj, (size_t)nb * pl * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_pl, d + o_pl, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_lx, lx, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_ly, ly, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_c, d + o_c, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipEventRecord(c.done[slot], st));
    pend[slot].active = true;
    pend[slot].b0 = b0;
    pend[slot].nb = nb;
  }
  if (int rc = finish(chunk & 1)) return rc;
  if (int rc = finish((chunk + 1) & 1)) return rc;
  return 0;
}

}  // extern "C"
