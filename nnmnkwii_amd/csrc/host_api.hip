// Host-pointer entry points: numpy-in / numpy-out calls without any framework tensor in between.
//
// mlpg_hip_forward_host replaces the loop a user of the reference writes around paramgen.mlpg for a padded batch
// held in HOST memory (util/__init__.py:44-66 over datasets/__init__.py:152-218 arrays).  The batch is cut into
// chunks of utterances that alternate between two HIP streams; per chunk: host -> device copy, the MLPG kernels,
// device -> host copy, all asynchronous, so that the PCIe transfers of one chunk run under the kernels of the other
// and under the CPU-side staging of the next.  Pageable host memory is staged through pinned buffers by a few copy
// threads (a single-threaded memcpy is slower than PCIe Gen5); memory that is already pinned (mlpg_hip_host_alloc,
// hipHostMalloc, hipHostRegister) is transferred in place.
//
// The *_multi forms take a LIST of devices (utterances / pairs are independent, so a batch shards
// with no exchange at all): chunk c goes to list entry c % n, on that entry's stream pair slot (c / n) % 2
// (host_chunk_plan below -- the whole dealing logic, exported for the CPU tests as mlpg_hip_host_chunk_plan).  One host
// thread drives every device; per device the behaviour is exactly the single-device one, and results land in the
// caller's arrays at the chunk's own offset, so the merge of outputs and verdicts is positional.  A device may appear
// several times in the list (each occurrence gets its own staging context): that is how the GPU tests exercise the
// dealing on a one-GPU box.
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
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
constexpr int kMaxHostDevices = 16;  // device indices the host entry points accept
constexpr int kMaxRep = 4;           // staging contexts per device (occurrences of one device in a device list)
constexpr int kMaxList = 32;         // entries of a device list
HostCtx g_host[kMaxHostDevices][kMaxRep];
std::mutex g_host_mu;
std::atomic<long long> g_host_chunks[kMaxHostDevices];  // chunks the *_host_multi calls enqueued per device (a test aid)

// A resolved device list: entry k runs on device dev[k] with staging context ctx[k].
struct DevList {
  int n = 0;
  int dev[kMaxList];
  HostCtx *ctx[kMaxList];
};

// devices == NULL or num_devices <= 0: every visible device once.
int resolve_devices(const int32_t *devices, int num_devices, DevList *dl) {
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess) {
    (void)hipGetLastError();
    visible = 0;
  }
  if (visible <= 0) {
    set_error("no HIP device is visible");
    return MLPG_HIP_EINVAL;
  }
  if (visible > kMaxHostDevices) visible = kMaxHostDevices;
  int used[kMaxHostDevices] = {};
  dl->n = 0;
  if (!devices || num_devices <= 0) {
    for (int d = 0; d < visible; ++d) {
      dl->dev[dl->n] = d;
      dl->ctx[dl->n++] = &g_host[d][0];
    }
    return 0;
  }
  if (num_devices > kMaxList) {
    set_error("device list longer than %d", kMaxList);
    return MLPG_HIP_EINVAL;
  }
  for (int k = 0; k < num_devices; ++k) {
    const int d = devices[k];
    if (d < 0 || d >= visible) {
      set_error("bad device %d (%d visible)", d, visible);
      return MLPG_HIP_EINVAL;
    }
    if (used[d] >= kMaxRep) {
      set_error("device %d appears more than %d times in the device list", d, kMaxRep);
      return MLPG_HIP_EINVAL;
    }
    dl->dev[dl->n] = d;
    dl->ctx[dl->n++] = &g_host[d][used[d]++];
  }
  return 0;
}

// Chunk c of a batch cut into chunks of `chunk` items, dealt over n list entries.
inline void host_chunk_owner(long c, int n, int *entry, int *slot) {
  *entry = (int)(c % n);
  *slot = (int)((c / n) & 1);
}

// Items per chunk: about `target` items, but at least 4 chunks per device when the batch allows (so that the transfers of
// one chunk run under the kernels of another on every device).
inline long host_chunk_items(long n_items, long target, int n) {
  return std::max<long>(1, std::min<long>(target, (n_items + 4L * n - 1) / (4L * n)));
}

// Populate the page tables of a pageable output array in the background (its first touch otherwise happens inside the
// final host copy: 6 ms of page faults for config 2's 123 MB).  MADV_POPULATE_WRITE leaves the content alone, so it
// needs no ordering against the copies; where the kernel does not know it nothing happens.
struct Prefault {
  std::thread th;
  void start(void *p, size_t bytes) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if (bytes < (8u << 20) || page == 0) return;
    const uintptr_t lo = ((uintptr_t)p + page - 1) & ~(uintptr_t)(page - 1), hi = ((uintptr_t)p + bytes) & ~(uintptr_t)(page - 1);
    if (hi <= lo) return;
    th = std::thread([lo, hi] {
      const size_t step = 8u << 20;
      for (uintptr_t a = lo; a < hi; a += step)
        if (madvise((void *)a, std::min<size_t>(step, hi - a), 23 /* MADV_POPULATE_WRITE */) != 0) return;
    });
  }
  ~Prefault() { if (th.joinable()) th.join(); }
};

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
  static const unsigned max_threads = [] {
    const char *e = getenv("MLPG_HIP_HOST_COPY_THREADS");
    const long v = e ? atol(e) : 0;
    return (unsigned)(v > 0 ? std::min<long>(v, 64) : 16);
  }();
  unsigned nt = std::min<unsigned>(max_threads, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
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

// ---- the short path: ONE small call (round 6) -------------------------------------------------------------------
// The literal drop-in call is paramgen.mlpg(mean_frames (T, D), variance_frames, windows) on numpy arrays, once per
// utterance (paramgen/_mlpg.py:92; the loop around it: util/__init__.py:56-66): 9.6 KB in at BASELINE config 1, 2.9 MB
// at one config-2 utterance.  At those sizes the chunk plan above is all overhead (a collector thread is created and
// joined per call: 30 us; five copies and an event per chunk; the reference itself needs 40 us at config 1).  The short
// path has one stream, one pinned staging buffer and no thread (tools/dbg/small_call_latency.hip measured the pieces):
//   * inputs go pageable -> pinned by memcpy, means first, and each array's host -> device copy is enqueued as soon as
//     it is staged, so the variances are staged under the means' transfer; a batch of at most kSmallDirectIn bytes is
//     not copied to the device at all -- the kernel reads the pinned buffer over PCIe (one copy-engine hop, 3 us, less);
//   * the kernel writes trajectory and verdicts straight into pinned host memory (no device -> host copy: 10-12 us less
//     at 0.5 MB), followed by a one-thread kernel that writes a sequence number behind them;
//   * the host polls that word instead of hipStreamSynchronize (6.3 against 10.9 us for an empty launch), looking at
//     hipStreamQuery every few thousand polls so that a failed launch is an error, not a hang.
//   * an array of at least MLPG_HIP_HOST_DIRECT_KB (default 1200) is not staged at all: hipMemcpyAsync takes it straight from (the
//     result: to) the caller's pageable memory at the pinned rate (host_small below; tools/dbg/pageable_direct.hip).
// MLPG_HIP_HOST_SMALL_MB: calls of at most this many MB of input take the short path (default 64 -- up to there one stream and
// the runtime's direct copies beat the chunk plan's threads by 1.6-1.9 x, profiles/r06_host_direct_ab.txt; 0: never).
// The arrays (address, size) of the last calls that were large enough for a direct copy: see host_small.
struct SeenArrays {
  static constexpr int kN = 16;
  const void *ptr[kN] = {};
  size_t bytes[kN] = {};
  int next = 0;
  // was (p, n) handed in by one of the last calls?  remembers it either way
  bool seen_and_note(const void *p, size_t n) {
    for (int k = 0; k < kN; ++k)
      if (ptr[k] == p && bytes[k] == n) return true;
    ptr[next] = p;
    bytes[next] = n;
    next = (next + 1) % kN;
    return false;
  }
};
struct SmallCtx {
  SeenArrays seen;
  hipStream_t st = nullptr;
  char *pin_in = nullptr, *pin_out = nullptr, *dev = nullptr;
  size_t pin_in_bytes = 0, pin_out_bytes = 0, dev_bytes = 0;
  unsigned *flag = nullptr;  // one pinned cache line
  unsigned seq = 0;
};
SmallCtx g_small[kMaxHostDevices];
constexpr size_t kSmallDirectIn = 48u << 10;

__global__ void small_flag_kernel(volatile unsigned *flag, unsigned v) {
  *flag = v;
  __threadfence_system();
}

size_t small_limit_bytes() {
  static const size_t lim = [] {
    const char *e = getenv("MLPG_HIP_HOST_SMALL_MB");
    const double v = e ? atof(e) : 64.0;
    return (size_t)(v <= 0 ? 0 : v * 1048576.0);
  }();
  return lim;
}
// MLPG_HIP_HOST_DIRECT_KB: an array of at least this many KB that the library has been handed before (same address and size, one of
// the last 16) is copied by the runtime straight from / to the caller's memory instead of through the pinned staging buffers
// (default 1200; 0: never); MLPG_HIP_HOST_DIRECT_ALWAYS_KB: from this size on also an array seen for the first time (default 4096).
size_t host_direct_bytes() {
  static const size_t thr = [] {
    const char *e = getenv("MLPG_HIP_HOST_DIRECT_KB");
    const double v = e ? atof(e) : 1200.0;
    return (size_t)(v <= 0 ? 0 : v * 1024.0);
  }();
  return thr;
}
size_t host_direct_always_bytes() {
  static const size_t thr = [] {
    const char *e = getenv("MLPG_HIP_HOST_DIRECT_ALWAYS_KB");
    const double v = e ? atof(e) : 4096.0;
    return (size_t)(v <= 0 ? 0 : v * 1024.0);
  }();
  return thr;
}
// MLPG_HIP_HOST_PREFAULT=0: do not touch the result array's pages while the device works (A/B runs)
bool host_prefault() {
  static const bool on = [] { const char *e = getenv("MLPG_HIP_HOST_PREFAULT"); return !(e && e[0] == '0'); }();
  return on;
}
// MLPG_HIP_HOST_SMALL_WAIT=sync: hipStreamSynchronize instead of polling the flag word (A/B runs)
bool small_wait_by_flag() {
  static const bool flag = [] { const char *e = getenv("MLPG_HIP_HOST_SMALL_WAIT"); return !(e && e[0] == 's'); }();
  return flag;
}

// Staging helpers of the short path: a few threads that share the pageable <-> pinned copies of one call with the calling
// thread (a config-2 utterance: 2.9 MB in, 58 us for one core; the host -> device transfer behind it takes 60 us, so every
// microsecond of staging in front of the first transfer is on the call's critical path).  Not created per call (thread
// creation is 30 us): they spin on the job word for MLPG_HIP_HOST_SPIN_US microseconds after their last job (default 250: a
// loop of per-utterance calls keeps them awake) and then sleep on a condition variable.  A job is a list of 64 KB slices
// claimed through ONE 64-bit word (generation << 32 | next slice), so a helper that wakes up late cannot claim a slice of a
// job that is already over.  MLPG_HIP_HOST_HELPERS: number of helpers (default 3, at most half the cores - 1; 0: none).
class CopyPool {
 public:
  static CopyPool *get() {
    std::lock_guard<std::mutex> lk(inst_mu_);
    if (!inst_) {
      const char *e = getenv("MLPG_HIP_HOST_HELPERS");
      long n = e ? atol(e) : 3;
      const long cap = (long)std::thread::hardware_concurrency() / 2 - 1;
      n = std::max<long>(0, std::min<long>(std::min<long>(n, cap), kMaxHelpers));
      const char *s = getenv("MLPG_HIP_HOST_SPIN_US");
      inst_ = new CopyPool((int)n, s ? atof(s) : 250.0);
    }
    return inst_;
  }
  static void shutdown() {
    CopyPool *p = nullptr;
    {
      std::lock_guard<std::mutex> lk(inst_mu_);
      p = inst_;
      inst_ = nullptr;
    }
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(p->mu_);
      p->quit_.store(true);
    }
    p->cv_.notify_all();
    for (auto &t : p->th_) t.join();
    delete p;
  }
  // dst[0, bytes) = src[0, bytes), shared with the helpers; returns when every byte is in place
  void copy(void *dst, const void *src, size_t bytes) {
    if (th_.empty() || bytes < 2 * kSlice) {
      memcpy(dst, src, bytes);
      return;
    }
    const unsigned nslices = (unsigned)((bytes + kSlice - 1) / kSlice);
    dst_.store((char *)dst, std::memory_order_relaxed);
    src_.store((const char *)src, std::memory_order_relaxed);
    bytes_.store(bytes, std::memory_order_relaxed);
    nslices_.store(nslices, std::memory_order_relaxed);
    done_.store(0, std::memory_order_relaxed);
    const uint64_t g = (uint64_t)(++gen_) << 32;
    ticket_.store(g, std::memory_order_seq_cst);  // the job is published
    if (sleepers_.load(std::memory_order_seq_cst) > 0) {
      { std::lock_guard<std::mutex> lk(mu_); }
      cv_.notify_all();
    }
    work(gen_);
    while (done_.load(std::memory_order_acquire) != nslices) __builtin_ia32_pause();
  }

 private:
  static constexpr size_t kSlice = 64u << 10;
  static constexpr int kMaxHelpers = 7;
  CopyPool(int n, double spin_us) : spin_us_(spin_us) {
    for (int k = 0; k < n; ++k) th_.emplace_back([this] { helper(); });
  }
  // claims and copies slices of job `gen` until none is left (or the job is over)
  void work(unsigned gen) {
    for (;;) {
      uint64_t t = ticket_.load(std::memory_order_acquire);
      if ((unsigned)(t >> 32) != gen) return;
      const unsigned idx = (unsigned)t;
      if (idx >= nslices_.load(std::memory_order_relaxed)) return;
      if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      // the claim succeeded on a ticket of generation `gen`: the job fields are that job's (the caller does not publish the
      // next one before every claimed slice is counted in done_)
      const size_t off = (size_t)idx * kSlice;
      memcpy(dst_.load(std::memory_order_relaxed) + off, src_.load(std::memory_order_relaxed) + off,
             std::min(kSlice, bytes_.load(std::memory_order_relaxed) - off));
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  void helper() {
    unsigned seen = 0;
    auto last = std::chrono::steady_clock::now();
    for (unsigned long spins = 0;; ++spins) {
      const unsigned g = (unsigned)(ticket_.load(std::memory_order_acquire) >> 32);
      if (g != seen) {
        seen = g;
        work(g);
        last = std::chrono::steady_clock::now();
        continue;
      }
      if (quit_.load(std::memory_order_relaxed)) return;
      __builtin_ia32_pause();
      if ((spins & 255) == 255 &&
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - last).count() > spin_us_) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1, std::memory_order_seq_cst);
        cv_.wait(lk, [&] { return quit_.load() || (unsigned)(ticket_.load(std::memory_order_seq_cst) >> 32) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_seq_cst);
        if (quit_.load()) return;
        last = std::chrono::steady_clock::now();
      }
    }
  }
  static CopyPool *inst_;
  static std::mutex inst_mu_;
  std::vector<std::thread> th_;
  double spin_us_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<bool> quit_{false};
  std::atomic<int> sleepers_{0};
  std::atomic<uint64_t> ticket_{0};
  std::atomic<unsigned> done_{0};
  unsigned gen_ = 0;
  std::atomic<unsigned> nslices_{0};
  std::atomic<char *> dst_{nullptr};
  std::atomic<const char *> src_{nullptr};
  std::atomic<size_t> bytes_{0};
};
CopyPool *CopyPool::inst_ = nullptr;
std::mutex CopyPool::inst_mu_;

int small_ensure(SmallCtx &c, size_t in_bytes, size_t out_bytes, size_t dev_bytes) {
  if (!c.st) {
    MLPG_HIP_CHECK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    MLPG_HIP_CHECK(hipHostMalloc((void **)&c.flag, 64, hipHostMallocDefault));
    *c.flag = 0;
  }
  auto grow = [&](char *&buf, size_t &have, size_t want, bool host) -> int {
    if (have >= want) return 0;
    MLPG_HIP_CHECK(hipStreamSynchronize(c.st));
    if (buf) (void)(host ? hipHostFree(buf) : hipFree(buf));
    buf = nullptr;
    have = 0;
    const size_t sz = std::max<size_t>(want + want / 2, 256u << 10);
    if (host) MLPG_HIP_CHECK(hipHostMalloc((void **)&buf, sz, hipHostMallocDefault));
    else MLPG_HIP_CHECK(hipMalloc((void **)&buf, sz));
    have = sz;
    return 0;
  };
  if (int rc = grow(c.pin_in, c.pin_in_bytes, in_bytes, true)) return rc;
  if (int rc = grow(c.pin_out, c.pin_out_bytes, out_bytes, true)) return rc;
  return grow(c.dev, c.dev_bytes, dev_bytes, false);
}

// Waits for everything enqueued on c.st: a sequence number written to pinned memory behind it, polled.
int small_wait(SmallCtx &c, bool by_sync = false) {
  if (by_sync || !small_wait_by_flag()) {
    MLPG_HIP_CHECK(hipStreamSynchronize(c.st));
    return 0;
  }
  const unsigned v = ++c.seq;
  hipLaunchKernelGGL(small_flag_kernel, dim3(1), dim3(1), 0, c.st, (volatile unsigned *)c.flag, v);
  MLPG_HIP_CHECK(hipGetLastError());
  volatile unsigned *f = c.flag;
  for (unsigned long spins = 1;; ++spins) {
    if (*f == v) return 0;
    __builtin_ia32_pause();
    if ((spins & 0x3fff) == 0) {  // (every ~16 k polls: a few hundred microseconds)
      const hipError_t q = hipStreamQuery(c.st);
      if (q == hipSuccess) return 0;  // the stream has drained: the word is there
      if (q != hipErrorNotReady) {
        set_error("host call: the stream failed: %s", hipGetErrorString(q));
        (void)hipGetLastError();
        return MLPG_HIP_ERUNTIME;
      }
    }
  }
}

__global__ void widen_f32(const float *__restrict__ src, double *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

}  // namespace

long long host_chunks_on_device(int device) { return device >= 0 && device < kMaxHostDevices ? g_host_chunks[device].load() : -1; }

void host_api_shutdown() {
  std::lock_guard<std::mutex> lk(g_host_mu);
  CopyPool::shutdown();
  for (int d = 0; d < kMaxHostDevices; ++d) {
    SmallCtx &c = g_small[d];
    if (!c.st) continue;
    (void)hipSetDevice(d);
    (void)hipStreamSynchronize(c.st);
    (void)hipStreamDestroy(c.st);
    if (c.flag) (void)hipHostFree(c.flag);
    if (c.pin_in) (void)hipHostFree(c.pin_in);
    if (c.pin_out) (void)hipHostFree(c.pin_out);
    if (c.dev) (void)hipFree(c.dev);
    c = SmallCtx();
  }
  for (int dr = 0; dr < kMaxHostDevices * kMaxRep; ++dr) {
    const int d = dr / kMaxRep;
    HostCtx &c = g_host[d][dr % kMaxRep];
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

extern "C" const char *mlpg_hip_last_error(void);

namespace {

// MLPG_HIP_HOST_TRACE=1: where the host thread of a host-memory call spends its time (stderr, one line per call).
struct HostTrace {
  bool on;
  double t_wait = 0, t_submit = 0, t_collect = 0, t0;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  HostTrace() {
    static const bool env = [] { const char *e = getenv("MLPG_HIP_HOST_TRACE"); return e && atoi(e) > 0; }();
    on = env;
    t0 = on ? now() : 0.0;
  }
  void report(long nchunks, int n) const {
    if (on)
      fprintf(stderr, "[mlpg_hip host] %ld chunks over %d list entries: %.2f ms total = %.2f waiting for the collector + %.2f staging/enqueue; collector thread: %.2f copying results\n",
              nchunks, n, 1e3 * (now() - t0), 1e3 * t_wait, 1e3 * t_submit, 1e3 * t_collect);
  }
};

// Runs `nchunks` chunks over the device list: submit(entry, slot, c) enqueues chunk c on dl.ctx[entry]->st[slot] (the
// device is current), collect(entry, slot, c) hands the finished chunk's staged results to the caller (on the collector
// thread, in chunk order).  A slot is collected before it is reused and everything is collected before the call
// returns.  After an error nothing is left in flight: every stream that was used is drained before it is returned.
template <class Submit, class Collect>
int run_chunks(const DevList &dl, long nchunks, Submit submit, Collect collect) {
  HostTrace tr;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) {
    (void)hipGetLastError();
    prev = -1;
  }
  if (nchunks == 1) {  // one chunk: nothing to overlap, so no collector thread (creating and joining one costs 30-50 us of a small call)
    int rc = 0;
    if (hipSetDevice(dl.dev[0]) != hipSuccess) {
      set_error("hipSetDevice(%d) failed: %s", dl.dev[0], hipGetErrorString(hipGetLastError()));
      rc = MLPG_HIP_ERUNTIME;
    }
    const double ts = tr.on ? HostTrace::now() : 0.0;
    if (!rc) {
      g_host_chunks[dl.dev[0]].fetch_add(1);
      rc = submit(0, 0, 0L);
    }
    if (tr.on) tr.t_submit += HostTrace::now() - ts;
    const double tw = tr.on ? HostTrace::now() : 0.0;
    if (!rc && hipStreamSynchronize(dl.ctx[0]->st[0]) != hipSuccess) {
      set_error("host call: waiting for the chunk on device %d failed: %s", dl.dev[0], hipGetErrorString(hipGetLastError()));
      rc = MLPG_HIP_ERUNTIME;
    }
    if (tr.on) tr.t_wait += HostTrace::now() - tw;
    const double tc = tr.on ? HostTrace::now() : 0.0;
    if (!rc) rc = collect(0, 0, 0L);
    if (tr.on) tr.t_collect = HostTrace::now() - tc;
    if (rc) {  // nothing is left in flight
      for (int k = 0; k < 2; ++k)
        if (dl.ctx[0]->st[k]) (void)hipStreamSynchronize(dl.ctx[0]->st[k]);
      (void)hipGetLastError();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    tr.report(nchunks, dl.n);
    return rc;
  }
  // The collector thread waits for each chunk's event (in chunk order) and moves its staged results into the caller's
  // arrays while this thread stages the next chunks: with pageable inputs the calling thread is the critical path.
  struct {
    std::mutex mu;
    std::condition_variable cv;
    long submitted = 0, collected = 0;
    int rc = 0;
    bool stop = false;
    double t_collect = 0;
    char msg[512] = "";  // the collector's error text (the library's last-error string is per thread)
  } col;
  std::thread collector([&] {
    for (long c = 0;; ++c) {
      {
        std::unique_lock<std::mutex> lk(col.mu);
        col.cv.wait(lk, [&] { return col.submitted > c || col.stop; });
        if (col.submitted <= c) return;
      }
      int e, slot;
      host_chunk_owner(c, dl.n, &e, &slot);
      int rc = 0;
      if (hipSetDevice(dl.dev[e]) != hipSuccess || hipEventSynchronize(dl.ctx[e]->done[slot]) != hipSuccess) {
        set_error("host call: waiting for chunk %ld on device %d failed: %s", c, dl.dev[e], hipGetErrorString(hipGetLastError()));
        rc = MLPG_HIP_ERUNTIME;
      }
      const double ta = tr.on ? HostTrace::now() : 0.0;
      if (!rc) rc = collect(e, slot, c);
      {
        std::lock_guard<std::mutex> lk(col.mu);
        col.collected = c + 1;
        if (rc && !col.rc) {
          col.rc = rc;
          snprintf(col.msg, sizeof(col.msg), "%s", mlpg_hip_last_error());
        }
        if (tr.on) col.t_collect += HostTrace::now() - ta;
      }
      col.cv.notify_all();
    }
  });
  auto wait_collected = [&](long c) -> int {  // chunk c's results are with the caller, its slot is free again
    const double ta = tr.on ? HostTrace::now() : 0.0;
    std::unique_lock<std::mutex> lk(col.mu);
    col.cv.wait(lk, [&] { return col.collected > c; });
    if (tr.on) tr.t_wait += HostTrace::now() - ta;
    if (col.rc) set_error("%s", col.msg);
    return col.rc;
  };
  int cur = -1, rc = 0;
  long done = 0;  // chunks handed to the collector
  for (long c = 0; c < nchunks && !rc; ++c) {
    int e, slot;
    host_chunk_owner(c, dl.n, &e, &slot);
    if (c >= 2L * dl.n && (rc = wait_collected(c - 2L * dl.n))) break;
    if (cur != dl.dev[e]) {
      if (hipSetDevice(dl.dev[e]) != hipSuccess) {
        set_error("hipSetDevice(%d) failed: %s", dl.dev[e], hipGetErrorString(hipGetLastError()));
        rc = MLPG_HIP_ERUNTIME;
        break;
      }
      cur = dl.dev[e];
    }
    const double ts = tr.on ? HostTrace::now() : 0.0;
    g_host_chunks[dl.dev[e]].fetch_add(1);
    if ((rc = submit(e, slot, c))) break;
    if (hipEventRecord(dl.ctx[e]->done[slot], dl.ctx[e]->st[slot]) != hipSuccess) {
      set_error("hipEventRecord failed: %s", hipGetErrorString(hipGetLastError()));
      rc = MLPG_HIP_ERUNTIME;
      break;
    }
    if (tr.on) tr.t_submit += HostTrace::now() - ts;
    {
      std::lock_guard<std::mutex> lk(col.mu);
      col.submitted = done = c + 1;
    }
    col.cv.notify_all();
  }
  if (done > 0) {
    const int rc2 = wait_collected(done - 1);  // everything that was enqueued is collected (also after an error)
    if (!rc) rc = rc2;
  }
  {
    std::lock_guard<std::mutex> lk(col.mu);
    col.stop = true;
  }
  col.cv.notify_all();
  collector.join();
  if (rc) {  // nothing is left in flight
    for (int e = 0; e < dl.n; ++e) {
      if (hipSetDevice(dl.dev[e]) != hipSuccess) continue;
      for (int k = 0; k < 2; ++k)
        if (dl.ctx[e]->st[k]) (void)hipStreamSynchronize(dl.ctx[e]->st[k]);
    }
    (void)hipGetLastError();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  tr.t_collect = col.t_collect;
  tr.report(nchunks, dl.n);
  return rc;
}

// One small MLPG call on one device (see SmallCtx), forward (first_h = the means (B, Tmax, D), out_h = the trajectories (B, Tmax, sd),
// both `dtype`) or backward (first_h = grad_out (B, Tmax, sd) of `dtype`, out_h = the gradient (B, Tmax, D) of `out_dtype`);
// g_host_mu is held by the caller.
int host_small(int device, int dtype, int out_dtype, int algo, bool backward, const void *first_h, const void *var_h, int var_mode,
               const int32_t *lengths_h, int B, int Tmax, int D, int num_windows, const WinSet &ws, void *out_h, int32_t *status_h) {
  HostTrace tr;
  int prev = -1;
  MLPG_HIP_CHECK(hipGetDevice(&prev));
  if (prev != device) MLPG_HIP_CHECK(hipSetDevice(device));
  struct Restore {
    int prev, dev;
    ~Restore() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
  } restore{prev, device};
  SmallCtx &c = g_small[device];
  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8, esz_out = out_dtype == MLPG_HIP_F32 ? 4 : 8;
  const int sd = D / num_windows;
  const bool fvar = var_mode == MLPG_HIP_VAR_FRAME;
  const size_t mean_bytes = (size_t)B * Tmax * (backward ? sd : D) * esz;  // (the first array: means, or grad_out in a backward call)
  const size_t out_bytes = (size_t)B * Tmax * (backward ? D : sd) * esz_out;
  const size_t var_bytes = fvar ? (size_t)B * Tmax * D * esz : (var_mode == MLPG_HIP_VAR_GLOBAL ? (size_t)D * esz : 0);
  const size_t len_bytes = lengths_h ? (size_t)B * sizeof(int32_t) : 0, st_bytes = (size_t)B * sd * sizeof(int32_t);
  // device layout: [first | variances | lengths | (the result, when it goes back by a copy)]
  const size_t o_var = up256(mean_bytes), o_len = o_var + up256(var_bytes), in_total = o_len + up256(len_bytes);
  const bool direct_in = in_total <= kSmallDirectIn;  // the kernel reads the pinned staging buffer itself
  // A large array is not staged: hipMemcpyAsync takes it straight from / to the caller's (pageable) memory -- the runtime maps the
  // pages for the copy engine, which then moves them at the pinned rate (tools/dbg/pageable_direct.hip: 1.44 MB 48-52 us against 45
  // from pinned memory; the call blocks the host for the copy's time), where staging costs a memcpy in front of the first piece and
  // a bubble per piece: a config-2 utterance 168 -> 131 us.  But the mapping of pages the device has never seen costs more than the
  // staging it saves up to a few MB (a list of 48 distinct T = 2000 utterances, one call each: 462 against 390 us; from 5.8 MB per
  // array on it is the other way round: 688 against 722-779 us), so the rule is: an array of at least host_direct_bytes() goes
  // direct if the library was handed the same address and size in one of its last calls (a caller's preallocated buffer, a loop
  // over a buffer that numpy's allocator hands out again), from host_direct_always_bytes() on also the first time
  // (profiles/r06_host_direct_ab*.txt).  Below ~1 MB the runtime itself stages, and the kernel writing into pinned memory + one
  // memcpy beats a device -> pageable copy (480 KB: 31 against 50 us).
  const size_t thr = host_direct_bytes(), thr_always = host_direct_always_bytes();
  auto via_runtime = [&](const void *ptr, size_t bytes) {
    if (!thr || bytes < thr || !ptr) return false;
    const bool seen = c.seen.seen_and_note(ptr, bytes);
    return seen || (thr_always && bytes >= thr_always);
  };
  const bool first_rt = !direct_in && via_runtime(first_h, mean_bytes), var_rt = !direct_in && via_runtime(var_h, var_bytes),
             out_rt = via_runtime(out_h, out_bytes);
  // pinned layout: the staged arrays only (the same offsets as on the device when nothing goes direct)
  size_t pin_need = 0;
  const size_t p_first = pin_need;
  if (!first_rt) pin_need += up256(mean_bytes);
  const size_t p_var = pin_need;
  if (!var_rt) pin_need += up256(var_bytes);
  const size_t p_len = pin_need;
  pin_need += up256(len_bytes);
  const size_t o_status = out_rt ? 0 : up256(out_bytes), out_total = o_status + up256(st_bytes);
  const size_t o_out_dev = direct_in ? 0 : in_total, dev_need = (direct_in ? 0 : in_total) + (out_rt ? up256(out_bytes) : 0);
  if (int rc = small_ensure(c, pin_need, out_total, dev_need)) return rc;
  note_launch(direct_in ? kCountHostSmallDirect : kCountHostSmall);
  char *src = direct_in ? c.pin_in : c.dev;
  CopyPool *pool = CopyPool::get();  // (small copies are plain memcpys of the calling thread)
  // Staged piece by piece (768 KB: the copy engine starts 15 us into the call instead of 30 at a config-2 utterance, and every
  // further transfer costs 3 us), each piece's transfer enqueued behind its staging: the next piece is staged under it.  (Enqueued
  // copies are submitted at once -- measured: a hipStreamSynchronize behind a host-side pause of the copy's length returns in
  // 0.6 us -- and two streams do not move a pair of halves faster than one moves them in a row.)
  // (A shorter very first piece -- 64 / 192 / 384 KB, so that the copy engine starts earlier -- was measured in round 6: no difference
  // beyond the 8 % a config-2 utterance's call varies from process to process.)
  auto stage_send = [&](size_t pin_off, size_t dev_off, const void *from, size_t bytes, size_t tail) -> int {  // tail: staged bytes right behind, sent along
    constexpr size_t kPiece = 768u << 10;
    for (size_t o = 0; o < bytes;) {
      const size_t n = std::min(kPiece, bytes - o);
      pool->copy(c.pin_in + pin_off + o, (const char *)from + o, n);
      const bool last = o + n >= bytes;
      if (!direct_in)
        MLPG_HIP_CHECK(hipMemcpyAsync(c.dev + dev_off + o, c.pin_in + pin_off + o, n + (last ? tail : 0), hipMemcpyHostToDevice, c.st));
      if (last) break;
      o += n;
    }
    return 0;
  };
  auto drained = [&](int rc) {  // an error return leaves nothing in flight on the cached staging buffers
    (void)hipStreamSynchronize(c.st);
    (void)hipGetLastError();
    return rc;
  };
  auto send_direct = [&](size_t dev_off, const void *from, size_t bytes) -> int {
    if (hipMemcpyAsync(c.dev + dev_off, from, bytes, hipMemcpyHostToDevice, c.st) != hipSuccess) {
      set_error("host call: copying %zu bytes from the caller's memory failed: %s", bytes, hipGetErrorString(hipGetLastError()));
      return MLPG_HIP_ERUNTIME;
    }
    return 0;
  };
  if (int rc = first_rt ? send_direct(0, first_h, mean_bytes) : stage_send(p_first, 0, first_h, mean_bytes, 0)) return drained(rc);
  if (len_bytes) memcpy(c.pin_in + p_len, lengths_h, len_bytes);
  // (staged variances and the lengths are neighbours in both buffers: the lengths ride on the variances' last transfer)
  bool len_sent = !len_bytes || direct_in;
  if (var_bytes) {
    if (var_rt) {
      if (int rc = send_direct(o_var, var_h, var_bytes)) return drained(rc);
    } else {
      if (int rc = stage_send(p_var, o_var, var_h, var_bytes, len_bytes ? o_len + len_bytes - (o_var + var_bytes) : 0)) return drained(rc);
      len_sent = true;
    }
  }
  if (!len_sent && hipMemcpyAsync(c.dev + o_len, c.pin_in + p_len, len_bytes, hipMemcpyHostToDevice, c.st) != hipSuccess) {
    set_error("host call: copying the lengths failed: %s", hipGetErrorString(hipGetLastError()));
    return drained(MLPG_HIP_ERUNTIME);
  }
  Problem p;
  p.mean = backward ? nullptr : src;
  p.var = var_bytes ? src + o_var : nullptr;
  p.grad_out = backward ? src : nullptr;
  p.lengths = len_bytes ? (const int32_t *)(src + o_len) : nullptr;
  p.out = out_rt ? c.dev + o_out_dev : c.pin_out;  // pinned host memory: written by the kernel over PCIe
  p.status = (int32_t *)(c.pin_out + o_status);
  p.var_mode = var_mode;
  p.B = B;
  p.Tmax = Tmax;
  p.D = D;
  p.sd = sd;
  p.ld_in = D;
  p.ld_gout = backward ? sd : 0;
  p.ld_out = backward ? D : sd;
  p.ld_status = sd;
  const double ts = tr.on ? HostTrace::now() : 0.0;
  int rc = dispatch_solve(c.st, dtype, out_dtype, algo, backward, p, ws, device);
  if (!rc && out_rt && hipMemcpyAsync(out_h, c.dev + o_out_dev, out_bytes, hipMemcpyDeviceToHost, c.st) != hipSuccess) {
    set_error("host call: copying %zu bytes into the caller's memory failed: %s", out_bytes, hipGetErrorString(hipGetLastError()));
    rc = MLPG_HIP_ERUNTIME;
  }
  // While the device works: first touch of the result array's pages.  A caller that keeps its results hands in a fresh array every
  // call (numpy's empty(): not one page of it exists yet), and the copy out would take its page faults one by one after the wait;
  // here they cost nothing (the wait is 90 us at a config-2 utterance, 120 pages take 40).  Pages that exist cost a store each.
  if (!rc && !out_rt && out_bytes >= (16u << 10) && host_prefault()) {
    volatile char *q = (volatile char *)out_h;
    for (size_t o = 0; o < out_bytes; o += 4096) q[o] = 0;
    q[out_bytes - 1] = 0;
  }
  // (a copy into the caller's pageable memory is the runtime's business down to its last host-side step: wait for it the runtime's way)
  if (!rc) rc = small_wait(c, out_rt);
  if (rc) return drained(rc);
  const double tw = tr.on ? HostTrace::now() : 0.0;
  if (!out_rt) pool->copy(out_h, c.pin_out, out_bytes);
  if (status_h) memcpy(status_h, c.pin_out + o_status, st_bytes);
  if (tr.on)
    fprintf(stderr, "[mlpg_hip host] short path (%s inputs%s%s%s): %.1f us total = %.1f staging/enqueue + %.1f launch/wait + %.1f copying results\n",
            direct_in ? "pinned, read by the kernel" : "copied to the device", first_rt ? ", first array straight from the caller's memory" : "",
            var_rt ? ", variances straight from the caller's memory" : "", out_rt ? ", result copied straight into the caller's memory" : "",
            1e6 * (HostTrace::now() - tr.t0), 1e6 * (ts - tr.t0), 1e6 * (tw - ts), 1e6 * (HostTrace::now() - tw));
  return 0;
}

int ensure_all(const DevList &dl, size_t in_bytes, size_t out_bytes, size_t dev_bytes) {
  int prev = -1;
  MLPG_HIP_CHECK(hipGetDevice(&prev));
  int rc = 0;
  for (int e = 0; e < dl.n && !rc; ++e) {
    if (hipSetDevice(dl.dev[e]) != hipSuccess) {
      set_error("hipSetDevice(%d) failed: %s", dl.dev[e], hipGetErrorString(hipGetLastError()));
      rc = MLPG_HIP_ERUNTIME;
      break;
    }
    rc = ensure(*dl.ctx[e], in_bytes, out_bytes, dev_bytes);
  }
  (void)hipSetDevice(prev);
  return rc;
}

}  // namespace

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

// Test aid: the staging copy of the short path (the calling thread and the helper threads share 64 KB slices); no GPU involved.
__attribute__((visibility("default"))) int mlpg_hip_host_copy(void *dst, const void *src, size_t bytes) {
  if (bytes && (!dst || !src)) {
    set_error("host_copy: NULL pointer");
    return MLPG_HIP_EINVAL;
  }
  std::lock_guard<std::mutex> lk(g_host_mu);  // (the pool serves one copy at a time: the host entry points hold this lock too)
  CopyPool::get()->copy(dst, src, bytes);
  return 0;
}

__attribute__((visibility("default"))) void mlpg_hip_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

__attribute__((visibility("default"))) long long mlpg_hip_host_chunk_plan(long long n_items, long long target_items,
                                                                         int num_devices, long long max_chunks,
                                                                         int32_t *entry, int32_t *slot, int64_t *first,
                                                                         int64_t *count) {
  if (n_items < 0 || target_items < 1 || num_devices < 1 || num_devices > kMaxList) {
    set_error("host_chunk_plan: bad arguments");
    return MLPG_HIP_EINVAL;
  }
  const long cb = host_chunk_items((long)n_items, (long)target_items, num_devices);
  const long long nchunks = (n_items + cb - 1) / cb;
  for (long long c = 0; c < nchunks && c < max_chunks; ++c) {
    int e, s_;
    host_chunk_owner((long)c, num_devices, &e, &s_);
    if (entry) entry[c] = e;
    if (slot) slot[c] = s_;
    if (first) first[c] = c * cb;
    if (count) count[c] = std::min<long long>(cb, n_items - c * cb);
  }
  return nchunks;
}

__attribute__((visibility("default"))) int mlpg_hip_forward_host_multi(const int32_t *devices, int num_devices, int dtype,
                                                                       int algo, const void *mean_h, const void *var_h,
                                                                       int var_mode, const int32_t *lengths_h, int B,
                                                                       int Tmax, int D, int num_windows,
                                                                       const int32_t *win_l_h, const int32_t *win_u_h,
                                                                       const double *win_coef_h, void *out_h,
                                                                       int32_t *status_h) {
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
  std::lock_guard<std::mutex> lk(g_host_mu);  // one host call at a time per process (the staging buffers are shared)
  DevList dl;
  if (int rc = resolve_devices(devices, num_devices, &dl)) return rc;
  const bool fvar = var_mode == MLPG_HIP_VAR_FRAME;
  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8;
  const int sd = D / num_windows;
  const size_t utt_in = (size_t)Tmax * D * esz, utt_out = (size_t)Tmax * sd * esz;
  // one small call on one device: the short path (no chunk plan, no thread, one stream)
  if (dl.n == 1 && (size_t)B * utt_in * (fvar ? 2 : 1) <= small_limit_bytes())
    return host_small(dl.dev[0], dtype, dtype, algo, false, mean_h, var_h, var_mode, lengths_h, B, Tmax, D, num_windows, ws, out_h, status_h);
  const bool mean_pinned = is_pinned(mean_h), var_pinned = fvar && is_pinned(var_h), out_pinned = is_pinned(out_h);

  // ~64 MB of input per chunk
  static const long chunk_mb = [] { const char *e = getenv("MLPG_HIP_HOST_CHUNK_MB"); const long v = e ? atol(e) : 0; return v > 0 ? v : 64; }();
  const long cb = host_chunk_items(B, (long)(((size_t)chunk_mb << 20) / (utt_in * (fvar ? 2 : 1))), dl.n);
  const long nchunks = (B + cb - 1) / cb;
  const size_t in_bytes = (size_t)cb * utt_in * (fvar ? 2 : 1);
  const size_t out_bytes = up256((size_t)cb * utt_out) + (size_t)cb * sd * sizeof(int32_t);
  const size_t o_mean = 0, o_var = up256((size_t)cb * utt_in), o_out = o_var + (fvar ? up256((size_t)cb * utt_in) : up256((size_t)D * esz)),
               o_status = o_out + up256((size_t)cb * utt_out), o_len = o_status + up256((size_t)cb * sd * sizeof(int32_t)),
               dev_bytes = o_len + up256((size_t)cb * sizeof(int32_t));
  // (Copying a chunk's pageable inputs straight from the caller's memory, as host_small does, instead of staging them by the copy
  // threads was measured in round 6, alternating inside one process on the whole config-2 batch: 25-28 ms either way,
  // profiles/r06_chunk_direct_ab.txt.)
  if (int rc = ensure_all(dl, in_bytes, out_bytes, dev_bytes)) return rc;
  Prefault prefault;
  if (!out_pinned) prefault.start(out_h, (size_t)B * utt_out);

  bool var_sent[kMaxList][2] = {};
  auto submit = [&](int e, int slot, long chunk) -> int {
    HostCtx &c = *dl.ctx[e];
    const long b0 = chunk * cb, nb = std::min<long>(cb, B - b0);
    hipStream_t st = c.st[slot];
    char *d = (char *)c.dev[slot];
    if (var_mode == MLPG_HIP_VAR_GLOBAL && !var_sent[e][slot]) {
      MLPG_HIP_CHECK(hipMemcpyAsync(d + o_var, var_h, (size_t)D * esz, hipMemcpyHostToDevice, st));
      var_sent[e][slot] = true;
    }
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
    if (int rc = dispatch_solve(st, dtype, dtype, algo, false, p, ws, dl.dev[e])) return rc;
    void *odst = out_pinned ? (void *)((char *)out_h + (size_t)b0 * utt_out) : c.pin_out[slot];
    MLPG_HIP_CHECK(hipMemcpyAsync(odst, d + o_out, (size_t)nb * utt_out, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync((char *)c.pin_out[slot] + up256((size_t)cb * utt_out), d + o_status,
                                  (size_t)nb * sd * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    return 0;
  };
  auto collect = [&](int e, int slot, long chunk) -> int {  // the chunk's staged results go to the caller's arrays
    HostCtx &c = *dl.ctx[e];
    const long b0 = chunk * cb, nb = std::min<long>(cb, B - b0);
    if (!out_pinned) parallel_copy((char *)out_h + (size_t)b0 * utt_out, c.pin_out[slot], (size_t)nb * utt_out);
    if (status_h) memcpy(status_h + (size_t)b0 * sd, (char *)c.pin_out[slot] + up256((size_t)cb * utt_out), (size_t)nb * sd * sizeof(int32_t));
    return 0;
  };
  return run_chunks(dl, nchunks, submit, collect);
}

__attribute__((visibility("default"))) int mlpg_hip_forward_host(int device, int dtype, int algo, const void *mean_h,
                                                                 const void *var_h, int var_mode,
                                                                 const int32_t *lengths_h, int B, int Tmax, int D,
                                                                 int num_windows, const int32_t *win_l_h,
                                                                 const int32_t *win_u_h, const double *win_coef_h,
                                                                 void *out_h, int32_t *status_h) {
  if (device < 0 || device >= kMaxHostDevices) {
    set_error("bad device %d", device);
    return MLPG_HIP_EINVAL;
  }
  const int32_t one = device;
  return mlpg_hip_forward_host_multi(&one, 1, dtype, algo, mean_h, var_h, var_mode, lengths_h, B, Tmax, D, num_windows, win_l_h,
                                     win_u_h, win_coef_h, out_h, status_h);
}

// MLPG backward on HOST memory: the literal paramgen.mlpg_grad call (paramgen/_mlpg.py:202-281: numpy in, float32 numpy out) and the
// backward of autograd.MLPG on CPU tensors (autograd/_impl/mlpg.py:57-67).  var_h / grad_out_h of in_dtype, grad_h (B, Tmax, D) of
// out_dtype, status_h (B * sd, may be NULL).  The batch goes through the short path of mlpg_hip_forward_host (one stream, pinned
// staging, the kernel writes the gradient into pinned host memory) in pieces of whole utterances of at most MLPG_HIP_HOST_SMALL_MB of
// input each, one after the other: the reference's call is one utterance; a large batch is better served by mlpg_hip_backward on
// device memory.
__attribute__((visibility("default"))) int mlpg_hip_backward_host(int device, int in_dtype, int out_dtype, int algo, const void *var_h,
                                                                  int var_mode, const void *grad_out_h, const int32_t *lengths_h,
                                                                  int B, int Tmax, int D, int num_windows, const int32_t *win_l_h,
                                                                  const int32_t *win_u_h, const double *win_coef_h, void *grad_h,
                                                                  int32_t *status_h) {
  if (device < 0 || device >= kMaxHostDevices) {
    set_error("bad device %d", device);
    return MLPG_HIP_EINVAL;
  }
  if (B < 0 || Tmax < 0 || D < 0 || num_windows < 1 || D % num_windows != 0) {
    set_error("backward_host: bad sizes (B=%d, Tmax=%d, D=%d, num_windows=%d)", B, Tmax, D, num_windows);
    return MLPG_HIP_EINVAL;
  }
  if ((in_dtype != MLPG_HIP_F32 && in_dtype != MLPG_HIP_F64) || (out_dtype != MLPG_HIP_F32 && out_dtype != MLPG_HIP_F64)) {
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
  if (!grad_out_h || !grad_h) {
    set_error("NULL data pointer");
    return MLPG_HIP_EINVAL;
  }
  std::lock_guard<std::mutex> lk(g_host_mu);
  const int32_t one = device;
  DevList dl;
  if (int rc = resolve_devices(&one, 1, &dl)) return rc;
  const bool fvar = var_mode == MLPG_HIP_VAR_FRAME;
  const size_t esz = in_dtype == MLPG_HIP_F32 ? 4 : 8, esz_out = out_dtype == MLPG_HIP_F32 ? 4 : 8;
  const int sd = D / num_windows;
  const size_t utt_go = (size_t)Tmax * sd * esz, utt_var = fvar ? (size_t)Tmax * D * esz : 0, utt_grad = (size_t)Tmax * D * esz_out;
  const size_t lim = std::max<size_t>(small_limit_bytes(), 1u << 20);
  const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, lim / (utt_go + utt_var)));
  for (int b0 = 0; b0 < B; b0 += per) {
    const int nb = std::min(per, B - b0);
    const char *v = !var_h ? nullptr : (fvar ? (const char *)var_h + (size_t)b0 * utt_var : (const char *)var_h);
    if (int rc = host_small(dl.dev[0], in_dtype, out_dtype, algo, true, (const char *)grad_out_h + (size_t)b0 * utt_go, v, var_mode,
                            lengths_h ? lengths_h + b0 : nullptr, nb, Tmax, D, num_windows, ws, (char *)grad_h + (size_t)b0 * utt_grad,
                            status_h ? status_h + (size_t)b0 * sd : nullptr))
      return rc;
  }
  return 0;
}

// fastdtw for N utterance pairs held in HOST memory: what DTWAligner.transform (preprocessing/alignment.py:41-76) does
// per pair -- trim_zeros_frames on both utterances (:46-47, when lenx_h / leny_h are NULL: trailing frames with
// sum |x| < eps are dropped on the device), fastdtw (:50) -- chunked over the pairs like mlpg_hip_forward_host, the
// transfers of one chunk under the kernels of the other.  X_h (N, Tx, D), Y_h (N, Ty, D) float32 or float64 (float32
// is widened on the device: the distances are computed in float64 either way).  Outputs as mlpg_hip_fastdtw;
// lenx_out_h / leny_out_h (may be NULL) receive the lengths that were used.
__attribute__((visibility("default"))) int mlpg_hip_fastdtw_host_multi(const int32_t *devices, int num_devices, int dtype,
                                                                       const void *X_h, const void *Y_h,
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
  std::lock_guard<std::mutex> lk(g_host_mu);
  DevList dl;
  if (int rc = resolve_devices(devices, num_devices, &dl)) return rc;

  const size_t esz = dtype == MLPG_HIP_F32 ? 4 : 8;
  const size_t px = (size_t)Tx * D, py = (size_t)Ty * D, pl = (size_t)Tx + Ty;  // elements per pair: X, Y, path slots
  const long cb = host_chunk_items(N, (long)((64u << 20) / ((px + py) * esz)), dl.n);
  const long nchunks = (N + cb - 1) / cb;
  // pinned staging: in = X | Y;  out = path_i | path_j | path_len | lenx | leny | cost
  const size_t in_bytes = up256((size_t)cb * px * esz) + (size_t)cb * py * esz;
  const size_t so_pj = up256((size_t)cb * pl * 4), so_pl = 2 * so_pj, so_lx = so_pl + up256((size_t)cb * 4),
               so_ly = so_lx + up256((size_t)cb * 4), so_c = so_ly + up256((size_t)cb * 4),
               out_bytes = so_c + (size_t)cb * 8;
  // device: X | Y | X64 | Y64 (float32 input only) | the results, laid out like the pinned block (a full chunk's results go back
  // in ONE copy: six small copies cost 8 us each behind a kernel that takes 300 us for one pair)
  const size_t o_x = 0, o_y = up256((size_t)cb * px * esz), o_x64 = o_y + up256((size_t)cb * py * esz),
               o_y64 = o_x64 + (dtype == MLPG_HIP_F32 ? up256((size_t)cb * px * 8) : 0),
               o_res = o_y64 + (dtype == MLPG_HIP_F32 ? up256((size_t)cb * py * 8) : 0), o_pi = o_res, o_pj = o_res + so_pj,
               o_pl = o_res + so_pl, o_lx = o_res + so_lx, o_ly = o_res + so_ly, o_c = o_res + so_c, dev_bytes = o_res + up256(out_bytes);
  if (int rc = ensure_all(dl, in_bytes, out_bytes, dev_bytes)) return rc;
  const bool x_pinned = is_pinned(X_h), y_pinned = is_pinned(Y_h);

  auto submit = [&](int e, int slot, long chunk) -> int {
    HostCtx &c = *dl.ctx[e];
    const long b0 = chunk * cb, nb = std::min<long>(cb, N - b0);
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
    if (int rc = launch_fastdtw(st, dl.dev[e], x64, y64, lx, ly, (int)nb, Tx, Ty, D, radius, dist_kind, dist_scale, tie_rule,
                                (int32_t *)(d + o_pi), (int32_t *)(d + o_pj), (int32_t *)(d + o_pl), (double *)(d + o_c)))
      return rc;
    char *o = (char *)c.pin_out[slot];
    if (nb == cb) {
      MLPG_HIP_CHECK(hipMemcpyAsync(o, d + o_res, out_bytes, hipMemcpyDeviceToHost, st));
      return 0;
    }
    MLPG_HIP_CHECK(hipMemcpyAsync(o, d + o_pi, (size_t)nb * pl * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_pj, d + o_pj, (size_t)nb * pl * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_pl, d + o_pl, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_lx, lx, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_ly, ly, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    MLPG_HIP_CHECK(hipMemcpyAsync(o + so_c, d + o_c, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    return 0;
  };
  auto collect = [&](int e, int slot, long chunk) -> int {
    const long b0 = chunk * cb, nb = std::min<long>(cb, N - b0);
    const char *o = (const char *)dl.ctx[e]->pin_out[slot];
    // (the path slots behind a pair's path are whatever the device buffer held: handed over as zeros, so that the arrays of
    // two identical calls are identical)
    const int32_t *plen = (const int32_t *)(o + so_pl);
    for (long n = 0; n < nb; ++n) {
      const size_t k = (size_t)std::min<long>(std::max<long>(plen[n], 0), (long)pl);
      int32_t *di = path_i_h + (size_t)(b0 + n) * pl, *dj = path_j_h + (size_t)(b0 + n) * pl;
      memcpy(di, o + (size_t)n * pl * 4, k * 4);
      memcpy(dj, o + so_pj + (size_t)n * pl * 4, k * 4);
      memset(di + k, 0, (pl - k) * 4);
      memset(dj + k, 0, (pl - k) * 4);
    }
    memcpy(path_len_h + b0, o + so_pl, (size_t)nb * 4);
    if (lenx_out_h) memcpy(lenx_out_h + b0, o + so_lx, (size_t)nb * 4);
    if (leny_out_h) memcpy(leny_out_h + b0, o + so_ly, (size_t)nb * 4);
    memcpy(cost_h + b0, o + so_c, (size_t)nb * 8);
    return 0;
  };
  return run_chunks(dl, nchunks, submit, collect);
}

__attribute__((visibility("default"))) int mlpg_hip_fastdtw_host(int device, int dtype, const void *X_h, const void *Y_h,
                                                                 const int32_t *lenx_h, const int32_t *leny_h, int N,
                                                                 int Tx, int Ty, int D, int radius, int dist_kind,
                                                                 double dist_scale, int tie_rule, double trim_eps,
                                                                 int32_t *path_i_h, int32_t *path_j_h,
                                                                 int32_t *path_len_h, double *cost_h,
                                                                 int32_t *lenx_out_h, int32_t *leny_out_h) {
  if (device < 0 || device >= kMaxHostDevices) {
    set_error("bad device %d", device);
    return MLPG_HIP_EINVAL;
  }
  const int32_t one = device;
  return mlpg_hip_fastdtw_host_multi(&one, 1, dtype, X_h, Y_h, lenx_h, leny_h, N, Tx, Ty, D, radius, dist_kind, dist_scale,
                                     tie_rule, trim_eps, path_i_h, path_j_h, path_len_h, cost_h, lenx_out_h, leny_out_h);
}

}  // extern "C"
