// Follow-up to stream_pattern.hip: does the strip kernel's ROW SHAPE cost bandwidth?
//   rows of 60 lanes x 8 B = 480 B, back to back (the 3 windows of a frame are 1440 contiguous bytes), 18 frames per
//   wavefront of which 2 are the neighbours' (halo), vs aligned 512-B rows, vs 16 B/lane; optionally with the
//   kernel's output stores (one 480-B row per frame) -- persistent grid of 512 workgroups, items in order.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: 480-B rows exactly as the kernel (lane < 60, 8 B), ring of RING frames (6 loads each) ; HALO: frames f0-1 .. f0+16
// MODE 1: the same bytes as aligned 512-B rows (64 lanes x 8 B), contiguous
// MODE 2: 16 B per lane rows of 1024 B, contiguous
template <int MODE, int RING, bool HALO, int STORE>
__global__ void __launch_bounds__(256) k(const char *__restrict__ a, const char *__restrict__ b, char *__restrict__ o,
                                         float *__restrict__ sink, int nutt, int T) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int R = (T + 63) / 64;
  const long nitems = (long)nutt * R;
  float acc = 0;
  for (long it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int u = (int)(it / R), r = (int)(it % R);
    const int f0 = (r * 4 + wv) * 16;
    const char *ua = a + (long)u * T * 1440, *ub = b + (long)u * T * 1440;
    if (MODE == 0) {
      const int first = HALO ? -1 : 0, n = HALO ? 18 : 16;
      const int ln = lane < 60 ? lane : 59;
      f2 va[RING][3], vb[RING][3];
      auto ld = [&](int slot, int i) {
        int t = f0 + first + i; t = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          va[slot][w] = *(const f2 *)(ua + (long)t * 1440 + w * 480 + ln * 8);
          vb[slot][w] = *(const f2 *)(ub + (long)t * 1440 + w * 480 + ln * 8);
        }
      };
#pragma unroll
      for (int i = 0; i < RING; ++i) ld(i, i);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        if (i < n) {
#pragma unroll
          for (int w = 0; w < 3; ++w) acc += va[i % RING][w].x * vb[i % RING][w].y;
          if (i + RING < n) ld(i % RING, i + RING);
        }
      }
    } else {
      // the chunk's 16 (or 18) frames x 1440 B as one contiguous run per array, read in aligned rows
      const long bytes = (HALO ? 18 : 16) * 1440L;
      long off = ((long)(f0 - (HALO ? 1 : 0)) * 1440) & ~1023L; if (off < 0) off = 0;
      const long rowb = MODE == 1 ? 512 : 1024;
      const long rows = (bytes + rowb - 1) / rowb;
      for (long q = 0; q < rows; q += 6) {
        float part = 0;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          const long rr = q + k2 < rows ? q + k2 : rows - 1;
          long p = off + rr * rowb; if (p + rowb > (long)T * 1440) p = (long)T * 1440 - rowb;
          if (MODE == 1) { part += (*(const f2 *)(ua + p + lane * 8)).x * (*(const f2 *)(ub + p + lane * 8)).y; }
          else { part += (*(const f4 *)(ua + p + lane * 16)).x * (*(const f4 *)(ub + p + lane * 16)).w; }
        }
        acc += part;
      }
    }
    if (STORE) {
      char *uo = o + (long)u * T * 480;
      if (STORE == 1 || STORE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = f0 + i;
          if (t < T && lane < 60) {
            f2 *p = (f2 *)(uo + (long)t * 480 + lane * 8);
            if (STORE == 2) __builtin_nontemporal_store(f2{acc, acc}, p); else *p = f2{acc, acc};
          }
        }
      } else {
        // rows i and i+1 in one instruction: even lanes row i, odd lanes row i+1, 16 B per lane (two adjacent dims)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int t = f0 + i + (lane & 1);
          if (t < T && lane < 60) {
            f4 *p = (f4 *)(uo + (long)t * 480 + (lane >> 1) * 16);
            if (STORE == 4) __builtin_nontemporal_store(f4{acc, acc, acc, acc}, p); else *p = f4{acc, acc, acc, acc};
          }
        }
      }
    }
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;
}

int main() {
  const int nutt = 256, T = 1000;
  const long nb = (long)nutt * T * 1440;
  char *a, *b, *o; float *s;
  if (hipMalloc(&a, nb + 4096) != hipSuccess || hipMalloc(&b, nb + 4096) != hipSuccess || hipMalloc(&o, nb / 3 + 4096) != hipSuccess || hipMalloc(&s, 1 << 20) != hipSuccess) return 1;
  (void)hipMemset(a, 1, nb); (void)hipMemset(b, 1, nb);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char *name, double bytes, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      (void)hipEventRecord(e0);
      for (int k2 = 0; k2 < 5; ++k2) launch();
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5; if (ms < best) best = ms;
    }
    printf("%-58s %.4f ms  algorithmic %.0f GB/s\n", name, best, bytes / best / 1e6);
  };
#define RUN(NAME, M, RG, H, S, G) run(NAME, (S ? 7.0 / 6.0 : 1.0) * 2.0 * nb, [&] { hipLaunchKernelGGL((k<M, RG, H, S>), dim3(G), dim3(256), 0, 0, a, b, o, s, nutt, T); })
  for (int G : {512, 256}) {
    printf("grid %d\n", G);
    RUN("480-B rows ring 6 no halo, no stores", 0, 6, false, 0, G);
    RUN("  + plain 480-B row stores", 0, 6, false, 1, G);
    RUN("  + nontemporal 480-B row stores", 0, 6, false, 2, G);
    RUN("  + paired rows, 16 B/lane plain", 0, 6, false, 3, G);
    RUN("  + paired rows, 16 B/lane nontemporal", 0, 6, false, 4, G);
  }
  return 0;
}
