// Can the short host path skip its pinned staging?  tools/dbg/small_call_latency.hip found hipMemcpy pageable -> device at 2.88 MB
// as fast as the pinned transfer (60.7 against 59.8 us) -- with the same source buffer every time.  Here: hipMemcpyAsync straight
// from pageable memory -- is the call asynchronous, what does it cost with a source that was never handed to the runtime before
// (64 buffers cycled; a buffer malloc'ed, written and freed per call, like a numpy temporary), two arrays back to back, and the whole
// call's shape (two input arrays in, a kernel writing n/6 to pinned memory, a polled flag) both ways.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/pageable_direct.hip -o tools/dbg/bin/pageable_direct -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
__global__ void flag_kernel(volatile int *flag, int v) {
  *flag = v;
  __threadfence_system();
}
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
  printf("  %-92s median %8.1f us   min %8.1f   p90 %8.1f\n", name, t[t.size() / 2], t[0], t[t.size() * 9 / 10]);
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t kMax = 32u << 20;
  char *pin_a, *pin_b, *dev_a, *dev_b;
  CK(hipHostMalloc((void **)&pin_a, kMax, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&pin_b, kMax, hipHostMallocDefault));
  CK(hipMalloc((void **)&dev_a, kMax));
  CK(hipMalloc((void **)&dev_b, kMax));
  memset(pin_a, 1, kMax);
  memset(pin_b, 2, kMax);
  std::vector<char *> page(64);
  for (auto &p : page) {
    p = (char *)malloc(kMax / 2 + 4096) + 64;  // (not page-aligned, like a numpy array's data)
    memset(p, 3, kMax / 2);
  }
  int *flag;
  CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
  *flag = 0;
  int tick = 0;
  auto wait_flag = [&] {
    ++tick;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, (volatile int *)flag, tick);
    while (*(volatile int *)flag != tick) {}
  };
  const size_t sizes[] = {65536, 262144, 480000, 720000, 1000000, 1440000, 2880000, 5760000, 11520000};
  for (size_t n : sizes) {
    printf("== %zu bytes per array\n", n);
    int k = 0;
    stat("hipMemcpyAsync pinned -> device + flag wait", 60, [&] {
      CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, st));
      wait_flag();
    });
    stat("hipMemcpyAsync PAGEABLE -> device: host time of the call alone (same source every time)", 60, [&] {
      CK(hipMemcpyAsync(dev_a, page[0], n, hipMemcpyHostToDevice, st));
    });
    CK(hipStreamSynchronize(st));
    stat("hipMemcpyAsync PAGEABLE -> device + flag wait, same source every time", 60, [&] {
      CK(hipMemcpyAsync(dev_a, page[0], n, hipMemcpyHostToDevice, st));
      wait_flag();
    });
    stat("hipMemcpyAsync PAGEABLE -> device + flag wait, 64 sources cycled", 128, [&] {
      CK(hipMemcpyAsync(dev_a, page[(k++) & 63], n, hipMemcpyHostToDevice, st));
      wait_flag();
    });
    stat("malloc + memset (not timed: included) ... see next line", 1, [&] {});
    {
      std::vector<double> t;
      for (int r = 0; r < 40; ++r) {
        char *p = (char *)malloc(n + 128);
        memset(p, r, n + 128);  // the producer wrote the array (pages present, never seen by the runtime)
        const double a = now_us();
        CK(hipMemcpyAsync(dev_a, p + 64, n, hipMemcpyHostToDevice, st));
        wait_flag();
        t.push_back(now_us() - a);
        free(p);
      }
      std::sort(t.begin(), t.end());
      printf("  %-92s median %8.1f us   min %8.1f   p90 %8.1f\n", "hipMemcpyAsync PAGEABLE -> device + flag wait, a fresh malloc per call", t[t.size() / 2], t[0],
             t[t.size() * 9 / 10]);
    }
    stat("TWO arrays PAGEABLE -> device back to back + flag wait, sources cycled", 64, [&] {
      CK(hipMemcpyAsync(dev_a, page[(k++) & 63], n, hipMemcpyHostToDevice, st));
      CK(hipMemcpyAsync(dev_a + kMax / 2, page[(k++) & 63], n, hipMemcpyHostToDevice, st));
      wait_flag();
    });
    const size_t n16 = (2 * n / 6 + 15) / 16;
    const int grid = (int)std::min<size_t>(1024, (n16 + 255) / 256);
    stat("the call's shape, DIRECT: two pageable arrays in, kernel device -> pinned (2n/6), flag, memcpy out", 64, [&] {
      CK(hipMemcpyAsync(dev_a, page[(k++) & 63], n, hipMemcpyHostToDevice, st));
      CK(hipMemcpyAsync(dev_a + kMax / 2, page[(k++) & 63], n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)pin_b, n16);
      wait_flag();
      memcpy(page[0], pin_b, n16 * 16);
    });
    stat("the call's shape, STAGED in 768 KB pieces (one thread): memcpy + pinned H2D pipelined, same tail", 64, [&] {
      const size_t piece = 768u << 10;
      for (int a = 0; a < 2; ++a) {
        const char *src = page[(k++) & 63];
        char *pin = pin_a + (size_t)a * (kMax / 2);
        for (size_t o = 0; o < n; o += piece) {
          const size_t m = std::min(piece, n - o);
          memcpy(pin + o, src + o, m);
          CK(hipMemcpyAsync(dev_a + (size_t)a * (kMax / 2) + o, pin + o, m, hipMemcpyHostToDevice, st));
        }
      }
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)pin_b, n16);
      wait_flag();
      memcpy(page[0], pin_b, n16 * 16);
    });
    stat("D2H: hipMemcpyAsync device -> PAGEABLE (n bytes) + flag wait, destinations cycled", 64, [&] {
      CK(hipMemcpyAsync(page[(k++) & 63], dev_a, n, hipMemcpyDeviceToHost, st));
      wait_flag();
    });
    stat("D2H: kernel device -> pinned (n bytes) + flag + memcpy pinned -> pageable (one thread)", 64, [&] {
      const size_t m16 = (n + 15) / 16;
      hipLaunchKernelGGL(copy16, dim3((int)std::min<size_t>(1024, (m16 + 255) / 256)), dim3(256), 0, st, (const uint4 *)dev_a, (uint4 *)pin_b, m16);
      wait_flag();
      memcpy(page[(k++) & 63], pin_b, n);
    });
  }
  return 0;
}
