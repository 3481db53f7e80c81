// Copy-kernel variants: which plain streaming copy reaches the highest HBM rate on this box?
// hipcc --offload-arch=gfx950 -O3 tools/dbg/copy_sweep.hip -o /tmp/copy_sweep && /tmp/copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT, int U>
__global__ void __launch_bounds__(256) k_stride(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * st < n; i += U * st) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * st) : s[i + u * st];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * st); else d[i + u * st] = v[u]; }
  }
  for (; i < n; i += st) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) k_oneshot(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = s[base + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) d[base + u * 256] = v[u];
}
template <int U>
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ s, float *__restrict__ out, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * st < n; i += U * st) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc += s[i + u * st];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
__global__ void __launch_bounds__(256) k_write(f4 *__restrict__ d, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  const f4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) d[i] = v;
}

int main() {
  const size_t nb = 512ull << 20, n = nb / 16;
  f4 *s, *d; float *o;
  CK(hipMalloc(&s, nb)); CK(hipMalloc(&d, nb)); CK(hipMalloc(&o, 1 << 20));
  CK(hipMemset(s, 1, nb)); CK(hipMemset(d, 0, nb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, double bytes, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      for (int k = 0; k < 10; ++k) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; if (ms < best) best = ms;
    }
    printf("%-44s %.4f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
  };
  for (int wg : {1024, 2048, 4096, 8192, 16384}) {
    char nm[96];
    snprintf(nm, 96, "stride nt  U4 grid %d", wg); run(nm, 2.0 * nb, [&] { hipLaunchKernelGGL((k_stride<true, 4>), dim3(wg), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, 96, "stride     U4 grid %d", wg); run(nm, 2.0 * nb, [&] { hipLaunchKernelGGL((k_stride<false, 4>), dim3(wg), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, 96, "stride     U8 grid %d", wg); run(nm, 2.0 * nb, [&] { hipLaunchKernelGGL((k_stride<false, 8>), dim3(wg), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, 96, "read-only  U8 grid %d", wg); run(nm, 1.0 * nb, [&] { hipLaunchKernelGGL((k_read<8>), dim3(wg), dim3(256), 0, 0, s, o, n); });
    snprintf(nm, 96, "write-only    grid %d", wg); run(nm, 1.0 * nb, [&] { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, d, n); });
  }
  run("oneshot U1", 2.0 * nb, [&] { hipLaunchKernelGGL((k_oneshot<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s, d, n); });
  run("oneshot U2", 2.0 * nb, [&] { hipLaunchKernelGGL((k_oneshot<2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, s, d, n); });
  run("oneshot U4", 2.0 * nb, [&] { hipLaunchKernelGGL((k_oneshot<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, s, d, n); });
  run("oneshot U8", 2.0 * nb, [&] { hipLaunchKernelGGL((k_oneshot<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, s, d, n); });
  run("hipMemcpyDtoD", 2.0 * nb, [&] { hipMemcpyAsync(d, s, nb, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
