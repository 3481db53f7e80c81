// micro-benchmark: dependent-chain latency of f64 VALU ops on gfx950 (cycles per op, one wave per CU / 8 waves per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_fma(double *o, double a, double b, int n, long long *cyc) {
  double x = o[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fma(x, a, b);
  }
  long long t1 = __builtin_readcyclecounter();
  o[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void chain_rcp(double *o, double a, double b, int n, long long *cyc) {
  double x = o[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_amdgcn_rcp(x) + a;
  }
  long long t1 = __builtin_readcyclecounter();
  o[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void chain_fma32(float *o, float a, float b, int n, long long *cyc) {
  float x = o[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, a, b);
  }
  long long t1 = __builtin_readcyclecounter();
  o[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *o; long long *c; hipMalloc(&o, 1 << 24); hipMalloc(&c, 1 << 16); hipMemset(o, 0, 1 << 24);
  long long h[8];
  const int n = 1000;
  for (int threads : {64, 256, 512}) {
    for (int blocks : {1, 512}) {
      hipLaunchKernelGGL(chain_fma, dim3(blocks), dim3(threads), 0, 0, o, 1.0000001, 1e-9, n, c); hipDeviceSynchronize();
      hipMemcpy(h, c, 8, hipMemcpyDeviceToHost); printf("fma_f64   threads %3d blocks %3d: %.1f cycles/op\n", threads, blocks, (double)h[0] / (16.0 * n));
      hipLaunchKernelGGL(chain_rcp, dim3(blocks), dim3(threads), 0, 0, o, 1.0000001, 1e-9, n, c); hipDeviceSynchronize();
      hipMemcpy(h, c, 8, hipMemcpyDeviceToHost); printf("rcp+add   threads %3d blocks %3d: %.1f cycles/pair\n", threads, blocks, (double)h[0] / (16.0 * n));
      hipLaunchKernelGGL(chain_fma32, dim3(blocks), dim3(threads), 0, 0, (float *)o, 1.0000001f, 1e-9f, n, c); hipDeviceSynchronize();
      hipMemcpy(h, c, 8, hipMemcpyDeviceToHost); printf("fma_f32   threads %3d blocks %3d: %.1f cycles/op\n", threads, blocks, (double)h[0] / (16.0 * n));
    }
  }
  return 0;
}
