// What does ONE small host-memory call cost on this box, piece by piece?  (round 6: the literal drop-in call
// paramgen.mlpg(numpy (T, D)) -> numpy goes through mlpg_hip_forward_host; BASELINE config 1 is 4.8 KB of means, one
// config-2 utterance 1.44 MB of means + 1.44 MB of variances in, 0.48 MB out.)
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/small_call_latency.hip -o tools/dbg/bin/small_call_latency -lpthread
// Measures, per size: single-thread memcpy pageable -> pinned (hot and cold source), hipMemcpyAsync pinned -> device (host time of
// the call, time to completion), hipMemcpy straight from pageable memory, an empty kernel launch + stream synchronize, a kernel
// reading / writing pinned host memory directly (no copy engine), device -> pinned copies, and the ways to wait for a stream
// (hipStreamSynchronize, polling hipStreamQuery, polling a flag the kernel writes to pinned memory).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void empty_kernel() {}
__global__ void flag_kernel(volatile int *flag, int v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *flag = v;
    __threadfence_system();
  }
}
// streaming copy, 16 B per lane: src / dst may be pinned host memory (read or written over PCIe by the kernel itself)
__global__ void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <class F>
static void stat(const char *name, int reps, F f) {
  std::vector<double> t;
  for (int r = 0; r < reps + 3; ++r) {
    const double a = now_us();
    f();
    const double b = now_us();
    if (r >= 3) t.push_back(b - a);
  }
  std::sort(t.begin(), t.end());
  printf("  %-78s median %8.1f us   min %8.1f   p90 %8.1f\n", name, t[t.size() / 2], t[0], t[t.size() * 9 / 10]);
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t kMax = 8u << 20;
  char *pin_a, *pin_b, *dev_a, *dev_b;
  CK(hipHostMalloc((void **)&pin_a, kMax, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&pin_b, kMax, hipHostMallocDefault));
  CK(hipMalloc((void **)&dev_a, kMax));
  CK(hipMalloc((void **)&dev_b, kMax));
  memset(pin_a, 1, kMax);
  memset(pin_b, 2, kMax);
  // 64 pageable source buffers (cycled: the "cold" source is one that was not touched for 63 calls)
  std::vector<char *> page(64);
  for (auto &p : page) {
    p = (char *)malloc(kMax);
    memset(p, 3, kMax);
  }
  int *flag;
  CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
  *flag = 0;

  printf("== fixed costs\n");
  stat("empty kernel launch (host time of the call)", 200, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); });
  CK(hipStreamSynchronize(st));
  stat("empty kernel launch + hipStreamSynchronize", 200, [&] {
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
    CK(hipStreamSynchronize(st));
  });
  stat("empty kernel launch + poll hipStreamQuery", 200, [&] {
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
    while (hipStreamQuery(st) == hipErrorNotReady) {}
  });
  int tick = 0;
  stat("flag kernel (writes pinned host word) + poll the word", 200, [&] {
    ++tick;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, st, (volatile int *)flag, tick);
    while (*(volatile int *)flag != tick) {}
  });
  CK(hipStreamSynchronize(st));
  stat("hipStreamSynchronize on an idle stream", 200, [&] { CK(hipStreamSynchronize(st)); });
  hipPointerAttribute_t attr;
  stat("hipPointerGetAttributes (pageable pointer)", 200, [&] {
    if (hipPointerGetAttributes(&attr, page[0]) != hipSuccess) (void)hipGetLastError();
  });
  stat("hipPointerGetAttributes (pinned pointer)", 200, [&] { CK(hipPointerGetAttributes(&attr, pin_a)); });
  int devq;
  stat("hipGetDevice + hipSetDevice(same)", 200, [&] {
    CK(hipGetDevice(&devq));
    CK(hipSetDevice(devq));
  });
  stat("std::thread create + join (empty)", 100, [&] {
    std::thread t([] {});
    t.join();
  });

  const size_t sizes[] = {4800, 9600, 65536, 480000, 1440000, 2880000, 5760000};
  for (size_t n : sizes) {
    const size_t n16 = (n + 15) / 16;
    printf("== %zu bytes\n", n);
    int k = 0;
    stat("memcpy pageable -> pinned, 1 thread, source cold (64 buffers cycled)", 120, [&] { memcpy(pin_a, page[(k++) & 63], n); });
    stat("memcpy pageable -> pinned, 1 thread, source hot (same buffer)", 120, [&] { memcpy(pin_a, page[0], n); });
    stat("memcpy pinned -> pageable, 1 thread", 120, [&] { memcpy(page[(k++) & 63], pin_b, n); });
    stat("hipMemcpyAsync pinned -> device: host time of the call", 100, [&] { CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st)); });
    CK(hipStreamSynchronize(st));
    stat("hipMemcpyAsync pinned -> device + hipStreamSynchronize", 100, [&] {
      CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
    });
    stat("hipMemcpyAsync device -> pinned + hipStreamSynchronize", 100, [&] {
      CK(hipMemcpyAsync(pin_b, dev_a, n, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    });
    stat("hipMemcpy pageable -> device (the runtime's own staging), source hot", 60, [&] { CK(hipMemcpy(dev_a, page[0], n, hipMemcpyHostToDevice)); });
    stat("hipMemcpy device -> pageable", 60, [&] { CK(hipMemcpy(page[1], dev_a, n, hipMemcpyDeviceToHost)); });
    const int grid = (int)std::min<size_t>(1024, (n16 + 255) / 256);
    stat("copy kernel pinned -> device (kernel reads host memory) + sync", 100, [&] {
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)pin_a, (uint4 *)dev_a, n16);
      CK(hipStreamSynchronize(st));
    });
    stat("copy kernel device -> pinned (kernel writes host memory) + sync", 100, [&] {
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)pin_b, n16);
      CK(hipStreamSynchronize(st));
    });
    stat("H2D copy, copy kernel device -> device, D2H copy of n/6, one sync (the call's shape)", 100, [&] {
      CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)dev_b, n16);
      CK(hipMemcpyAsync(pin_b, dev_b, (n / 6 + 15) & ~(size_t)15, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    });
    stat("H2D copy, kernel device -> PINNED n/6 (no D2H copy), one sync", 100, [&] {
      CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)pin_b, (n16 + 5) / 6);
      CK(hipStreamSynchronize(st));
    });
    // staged in 2 and 4 pieces: memcpy of piece k+1 under the transfer of piece k
    for (int pieces : {1, 2, 4}) {
      char nm[128];
      snprintf(nm, sizeof(nm), "memcpy (hot) + H2D in %d piece(s), pipelined, + sync", pieces);
      stat(nm, 60, [&] {
        const size_t per = ((n / pieces) + 255) & ~(size_t)255;
        for (size_t off = 0; off < n; off += per) {
          const size_t m = std::min(per, n - off);
          memcpy(pin_a + off, page[0] + off, m);
          CK(hipMemcpyAsync(dev_a + off, pin_a + off, m, hipMemcpyHostToDevice, st));
        }
        CK(hipStreamSynchronize(st));
      });
    }
    // is an enqueued copy submitted at once, or only with the next flush?  enqueue, spin 1.5 x its duration on the host
    // without touching the runtime, then synchronize: a sync that returns at once means the copy ran under the spin
    {
      const double est = 8.0 + n / 48e3;  // us
      double t_sync = 0;
      for (int r = 0; r < 20; ++r) {
        CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
        const double a = now_us();
        while (now_us() - a < 1.5 * est) {}
        const double b = now_us();
        CK(hipStreamSynchronize(st));
        if (r >= 4) t_sync += (now_us() - b) / 16;
      }
      printf("  %-78s %8.1f us (copy alone ~%.0f us)\n", "H2D enqueue, host spins 1.5 x the copy's time, THEN hipStreamSynchronize takes", t_sync, est);
      t_sync = 0;
      for (int r = 0; r < 20; ++r) {
        CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
        (void)hipStreamQuery(st);
        const double a = now_us();
        while (now_us() - a < 1.5 * est) {}
        const double b = now_us();
        CK(hipStreamSynchronize(st));
        if (r >= 4) t_sync += (now_us() - b) / 16;
      }
      printf("  %-78s %8.1f us\n", "   ... the same with a hipStreamQuery right behind the enqueue", t_sync);
    }
    stat("memcpy half, H2D, hipStreamQuery, memcpy half, H2D, sync", 60, [&] {
      const size_t h = (n / 2 + 255) & ~(size_t)255;
      memcpy(pin_a, page[0], std::min(h, n));
      CK(hipMemcpyAsync(dev_a, pin_a, std::min(h, n), hipMemcpyHostToDevice, st));
      (void)hipStreamQuery(st);
      if (n > h) {
        memcpy(pin_a + h, page[0] + h, n - h);
        CK(hipMemcpyAsync(dev_a + h, pin_a + h, n - h, hipMemcpyHostToDevice, st));
      }
      CK(hipStreamSynchronize(st));
    });
    stat("hipStreamQuery on a busy stream (host time of the call)", 60, [&] {
      CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
      const double a = now_us();
      (void)hipStreamQuery(st);
      const double b = now_us();
      CK(hipStreamSynchronize(st));
      (void)a; (void)b;
    });
    // copy kernel (shader reads pinned memory) for one half while the copy engine moves the other: two streams
    {
      static hipStream_t st2 = nullptr;
      if (!st2) CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
      const size_t h = ((n / 2) + 255) & ~(size_t)255;
      if (n > h) {
        stat("half by the copy engine (stream 1) + half by a copy kernel (stream 2), both synced", 60, [&] {
          CK(hipMemcpyAsync(dev_a, pin_a, h, hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(copy16, dim3(256), dim3(256), 0, st2, (const uint4 *)(pin_a + h), (uint4 *)(dev_a + h), (n - h) / 16);
          CK(hipStreamSynchronize(st));
          CK(hipStreamSynchronize(st2));
        });
        stat("two halves by the copy engine on two streams, both synced", 60, [&] {
          CK(hipMemcpyAsync(dev_a, pin_a, h, hipMemcpyHostToDevice, st));
          CK(hipMemcpyAsync(dev_a + h, pin_a + h, n - h, hipMemcpyHostToDevice, st2));
          CK(hipStreamSynchronize(st));
          CK(hipStreamSynchronize(st2));
        });
      }
    }
    // two threads staging halves (a helper spinning on an atomic: no thread creation inside the call)
    {
      std::atomic<int> go{0}, done{0};
      std::atomic<bool> quit{false};
      const char *src = page[0];
      std::thread helper([&] {
        int seen = 0;
        while (!quit.load(std::memory_order_relaxed)) {
          const int g = go.load(std::memory_order_acquire);
          if (g == seen) continue;
          seen = g;
          memcpy(pin_a + n / 2, src + n / 2, n - n / 2);
          done.store(g, std::memory_order_release);
        }
      });
      int gen = 0;
      stat("memcpy by 2 threads (spinning helper) + one H2D + sync", 60, [&] {
        ++gen;
        go.store(gen, std::memory_order_release);
        memcpy(pin_a, src, n / 2);
        while (done.load(std::memory_order_acquire) != gen) {}
        CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
      });
      quit.store(true);
      helper.join();
    }
  }
  return 0;
}
