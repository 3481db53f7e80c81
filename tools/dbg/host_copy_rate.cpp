// Rate of the staging copy of the host-memory entry points: pageable -> pinned (hipHostMalloc) and pageable -> malloc,
// by 1 .. 32 threads, 32 MB and 256 MB blocks.   hipcc -O2 -o host_copy_rate host_copy_rate.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static void pcopy(void *dst, const void *src, size_t bytes, unsigned nt) {
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

int main() {
  const size_t big = 768u << 20;
  char *src = (char *)malloc(big);
  memset(src, 1, big);
  for (size_t blk : {(size_t)32 << 20, (size_t)256 << 20}) {
    void *pin = nullptr;
    if (hipHostMalloc(&pin, blk, hipHostMallocDefault) != hipSuccess) return 1;
    char *pg = (char *)malloc(blk);
    memset(pg, 0, blk);
    memset(pin, 0, blk);
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u, 32u}) {
      for (int kind = 0; kind < 2; ++kind) {
        void *dst = kind ? (void *)pg : pin;
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
          const char *s = src + ((size_t)rep * blk) % (big - blk + 1);
          auto t0 = std::chrono::steady_clock::now();
          pcopy(dst, s, blk, nt);
          auto t1 = std::chrono::steady_clock::now();
          best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
        }
        printf("block %4zu MB  threads %2u  -> %-7s  %.2f ms  %.1f GB/s\n", blk >> 20, nt, kind ? "malloc" : "pinned", best * 1e3, blk / best / 1e9);
      }
    }
    hipHostFree(pin);
    free(pg);
  }
  return 0;
}
