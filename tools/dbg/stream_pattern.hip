// What HBM read rate does a strip-kernel-like access pattern reach, and what does it depend on?
// Workgroups of 4 wavefronts; an "item" is a contiguous block of C bytes in each of two arrays (means, variances); wavefront w
// reads quarter w of both blocks as rows of 64 lanes x W bytes with D row loads in flight.  Items are taken in order
// (item = blockIdx + k * gridDim: persistent) or one item per workgroup (one-shot: grid = number of items).
// hipcc --offload-arch=gfx950 -O3 tools/dbg/stream_pattern.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename V, int D>
__global__ void __launch_bounds__(256) k_items(const char *__restrict__ a, const char *__restrict__ b, float *__restrict__ out,
                                               long nitems, long C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long rowb = 64 * sizeof(V);
  const long q = C / 4;                 // bytes per wavefront and array
  const long rows = q / rowb;           // rows per wavefront and array
  V acc = {};
  for (long it = blockIdx.x; it < nitems; it += gridDim.x) {
    const char *pa = a + it * C + wv * q + lane * sizeof(V);
    const char *pb = b + it * C + wv * q + lane * sizeof(V);
    for (long r = 0; r < rows; r += D / 2) {
      V va[D / 2], vb[D / 2];
#pragma unroll
      for (int k = 0; k < D / 2; ++k) {
        const long rr = r + k < rows ? r + k : rows - 1;
        va[k] = *(const V *)(pa + rr * rowb);
        vb[k] = *(const V *)(pb + rr * rowb);
      }
#pragma unroll
      for (int k = 0; k < D / 2; ++k) acc += va[k] * vb[k];
    }
  }
  float s = 0;
  for (int k = 0; k < (int)(sizeof(V) / 4); ++k) s += acc[k];
  if (s == 12345.678f) out[blockIdx.x] = s;
}

int main() {
  const long nb = 368640000;  // one array of config 2 (256 x 1000 x 180 x 8)
  char *a, *b; float *o;
  if (hipMalloc(&a, nb + (1 << 20)) != hipSuccess || hipMalloc(&b, nb + (1 << 20)) != hipSuccess || hipMalloc(&o, 1 << 22) != hipSuccess) return 1;
  hipMemset(a, 1, nb); hipMemset(b, 1, nb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0);
      for (int k = 0; k < 5; ++k) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; if (ms < best) best = ms;
    }
    printf("%-60s %.4f ms  %.0f GB/s\n", name, best, 2.0 * nb / best / 1e6);
  };
  char nm[128];
  const long Cs[] = {92160, 46080, 23040, 184320};   // bytes per item and array: 64, 32, 16, 128 frames of 1440 B
  for (long C : Cs) {
    const long nitems = nb / C;
    for (int grid : {256, 512, 1024, 2048, 0}) {
      const int g = grid ? grid : (int)nitems;
      const char *gn = grid ? "persistent" : "one-shot  ";
      snprintf(nm, 128, "C=%6ld %s grid %6d  8B/lane D=12", C, gn, g); run(nm, [&] { hipLaunchKernelGGL((k_items<f2, 12>), dim3(g), dim3(256), 0, 0, a, b, o, nitems, C); });
      snprintf(nm, 128, "C=%6ld %s grid %6d  8B/lane D=36", C, gn, g); run(nm, [&] { hipLaunchKernelGGL((k_items<f2, 36>), dim3(g), dim3(256), 0, 0, a, b, o, nitems, C); });
      snprintf(nm, 128, "C=%6ld %s grid %6d 16B/lane D=6 ", C, gn, g); run(nm, [&] { hipLaunchKernelGGL((k_items<f4, 6>), dim3(g), dim3(256), 0, 0, a, b, o, nitems, C); });
      snprintf(nm, 128, "C=%6ld %s grid %6d 16B/lane D=18", C, gn, g); run(nm, [&] { hipLaunchKernelGGL((k_items<f4, 18>), dim3(g), dim3(256), 0, 0, a, b, o, nitems, C); });
    }
  }
  return 0;
}
