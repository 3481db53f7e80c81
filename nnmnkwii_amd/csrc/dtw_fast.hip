// fastdtw (Salvador & Chan 2007; semantics of slaypni/fastdtw's pure-Python
// implementation, restated in oracle/dtw_oracle.c) with the Euclidean local
// cost, one workgroup (one 64-lane wavefront) per utterance pair.
//
// Replaces the per-pair fastdtw(x, y, radius, dist) call of
// DTWAligner.transform (reference: preprocessing/alignment.py:50).
//
// Per pair:
//   1. halving pyramid of both series in an HBM scratch (pairwise means, odd
//      tail dropped), all levels;
//   2. from the coarsest level (full DTW) down to level 0: per-row windows
//      [lo_i, hi_i] from the coarser path (interval form of __expand_window:
//      the union of (2r+1)^2 neighbourhoods along a monotone path is one
//      interval per row), local costs for every window cell (all lanes in
//      parallel), then the DP recurrence swept along ANTI-DIAGONALS: lane r
//      owns row i0+r of a chunk of <= 64 rows and at step s handles column
//      s - r, so the cells done in one step are exactly one anti-diagonal;
//      rows hand values down through LDS; 1-byte back-pointers in LDS;
//   3. back-trace by one lane, new path kept in LDS for the next level.
// Everything except the pyramid lives in LDS (~42 KB for T = 900, radius 1).
//
// Arithmetic is bit-compatible with the oracle: cost = sqrt(sum_k (x-y)^2)
// with separate multiply/add in ascending k; D = min(up+dt, left+dt, diag+dt)
// compared after the add, first minimum wins (up, left, diagonal).
#include <math.h>

#include "common.h"

namespace mlpg {
namespace {

struct DtwParams {
  const double *X, *Y;
  const int32_t *lenx, *leny;
  int N, Tx, Ty, D, radius;
  int32_t *path_i, *path_j, *path_len;
  double *cost;
  double *pyr;        // N * pyr_stride doubles: x levels >= 1, then y levels >= 1
  size_t pyr_stride;  // (Tx + Ty) * D
  int cellcap;        // back-pointer capacity per level
  int chunkcap;       // D/cost cells per DP chunk
};

constexpr int kMaxLevels = 20;

__device__ __forceinline__ double l2_cost(const double *__restrict__ a, const double *__restrict__ b, int D) {
  double acc = 0.0;
  for (int k = 0; k < D; ++k) {
    const double diff = __dsub_rn(a[k], b[k]);
    acc = __dadd_rn(acc, __dmul_rn(diff, diff));
  }
  return __dsqrt_rn(acc);
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  *total = __shfl(incl, 63);
  return incl - v;
}

__global__ __launch_bounds__(64) void fastdtw_kernel(DtwParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int n = blockIdx.x;
  const int Tx = p.Tx, Ty = p.Ty, D = p.D, r = p.radius;
  const int pcap = Tx + Ty;

  // ---- LDS carve (doubles, ints, shorts, bytes) ----
  double *dchunk = (double *)smem;
  double *dprev = dchunk + p.chunkcap;
  int *off = (int *)(dprev + Ty);
  int *lvl_x = off + (Tx + 1);
  int *lvl_y = lvl_x + kMaxLevels;
  int *bcast = lvl_y + kMaxLevels;  // [4]
  unsigned short *lo = (unsigned short *)(bcast + 4);
  unsigned short *hi = lo + Tx;
  unsigned short *cfirst = hi + Tx;
  unsigned short *clast = cfirst + (Tx / 2 + 2);
  unsigned short *pth_i = clast + (Tx / 2 + 2);
  unsigned short *pth_j = pth_i + pcap;
  unsigned char *bp = (unsigned char *)(pth_j + pcap);

  const int tx = p.lenx[n], ty = p.leny[n];
  int32_t *out_i = p.path_i + (size_t)n * pcap;
  int32_t *out_j = p.path_j + (size_t)n * pcap;
  if (tx < 1 || ty < 1 || tx > Tx || ty > Ty) {
    if (lane == 0) {
      p.path_len[n] = 0;
      p.cost[n] = NAN;
    }
    return;
  }
  const double *x0 = p.X + (size_t)n * Tx * D;
  const double *y0 = p.Y + (size_t)n * Ty * D;
  double *px = p.pyr + (size_t)n * p.pyr_stride;
  double *py = px + (size_t)Tx * D;

  // number of halvings: level K is the first with a side < radius + 2 (full DTW there)
  int K = 0;
  while (K < kMaxLevels - 1 && (tx >> K) >= r + 2 && (ty >> K) >= r + 2) ++K;

  // ---- 1. pyramid (levels 1..K) ----
  if (lane == 0) {
    int xo = 0, yo = 0;
    for (int k = 1; k <= K; ++k) {
      lvl_x[k] = xo;
      lvl_y[k] = yo;
      xo += (tx >> k) * D;
      yo += (ty >> k) * D;
    }
  }
  __syncthreads();
  for (int side = 0; side < 2; ++side) {
    const double *src = side ? y0 : x0;
    double *base = side ? py : px;
    const int *lvl = side ? lvl_y : lvl_x;
    const int len0 = side ? ty : tx;
    for (int k = 1; k <= K; ++k) {
      double *dst = base + lvl[k];
      const int cnt = (len0 >> k) * D;
      for (int e = lane; e < cnt; e += 64) {
        const int row = e / D, c = e - row * D;
        dst[e] = __dadd_rn(src[(size_t)(2 * row) * D + c], src[(size_t)(2 * row + 1) * D + c]) * 0.5;
      }
      __threadfence_block();
      __syncthreads();
      src = dst;
    }
  }

  int pstart = pcap, pn = 0;  // current path = pth[pstart .. pstart+pn)
  double level_cost = INFINITY;
  bool fail = false;

  for (int k = K; k >= 0 && !fail; --k) {
    const int ltx = tx >> k, lty = ty >> k;
    const double *xk = k ? px + lvl_x[k] : x0;
    const double *yk = k ? py + lvl_y[k] : y0;

    // ---- 2a. per-row windows ----
    if (k == K) {
      for (int i = lane; i < ltx; i += 64) {
        lo[i] = 0;
        hi[i] = (unsigned short)(lty - 1);
      }
    } else {
      const int cx = tx >> (k + 1);
      for (int q = lane; q < pn; q += 64) {
        const int pi = pth_i[pstart + q], pj = pth_j[pstart + q];
        if (q == 0 || pth_i[pstart + q - 1] != pi) cfirst[pi] = (unsigned short)pj;
        if (q == pn - 1 || pth_i[pstart + q + 1] != pi) clast[pi] = (unsigned short)pj;
      }
      __syncthreads();
      for (int i = lane; i < ltx; i += 64) {
        const int ci = i >> 1;
        const int r0 = ci - r < 0 ? 0 : ci - r;
        const int r1 = ci + r > cx - 1 ? cx - 1 : ci + r;
        const int a = 2 * ((int)cfirst[r0] - r);
        const int b = 2 * ((int)clast[r1] + r) + 1;
        lo[i] = (unsigned short)(a < 0 ? 0 : a);
        hi[i] = (unsigned short)(b > lty - 1 ? lty - 1 : b);
      }
    }
    __syncthreads();

    // ---- 2b. row offsets (exclusive scan of the row widths) ----
    {
      const int rpl = (ltx + 63) / 64;
      const int b0 = lane * rpl < ltx ? lane * rpl : ltx;
      const int b1 = b0 + rpl < ltx ? b0 + rpl : ltx;
      int sum = 0;
      for (int i = b0; i < b1; ++i) sum += (int)hi[i] - (int)lo[i] + 1;
      int total;
      int run = wave_excl_scan(sum, lane, &total);
      for (int i = b0; i < b1; ++i) {
        off[i] = run;
        run += (int)hi[i] - (int)lo[i] + 1;
      }
      if (lane == 0) off[ltx] = total;
      if (total > p.cellcap) fail = true;
    }
    __syncthreads();
    if (fail) break;

    // ---- 2c. DP over chunks of rows ----
    int i0 = 0;
    int prevlo = 0, prevhi = -1;
    while (i0 < ltx) {
      const int base = off[i0];
      const bool fits = (i0 + lane < ltx) && (off[i0 + lane + 1] - base <= p.chunkcap);
      const unsigned long long m = __ballot(fits);
      const int R = (~m == 0ull) ? 64 : (__ffsll((long long)~m) - 1);
      if (R < 1) {
        fail = true;
        break;
      }
      const int ncell = off[i0 + R] - base;
      // local costs of every window cell of the chunk
      for (int c = lane; c < ncell; c += 64) {
        int a = 0, b = R;
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (off[i0 + mid] - base <= c) a = mid; else b = mid;
        }
        const int row = i0 + a;
        const int j = (int)lo[row] + c - (off[row] - base);
        dchunk[c] = l2_cost(xk + (size_t)row * D, yk + (size_t)j * D, D);
      }
      __syncthreads();
      // anti-diagonal sweep: lane = row, step s handles column s - lane
      const bool act = lane < R;
      const int i = i0 + lane;
      const int mylo = act ? (int)lo[i] : 0, myhi = act ? (int)hi[i] : -1;
      const int myoff = act ? off[i] - base : 0;
      const int gofs = act ? off[i] : 0;
      int uplo = prevlo, uphi = prevhi;
      const double *uprow = dprev;
      if (act && lane > 0) {
        uplo = (int)lo[i - 1];
        uphi = (int)hi[i - 1];
        uprow = dchunk + (off[i - 1] - base);
      }
      const int s0 = (int)lo[i0];
      const int s1 = (int)hi[i0 + R - 1] + R - 1;
      double left = INFINITY;
      for (int s = s0; s <= s1; ++s) {
        const int j = s - lane;
        if (act && j >= mylo && j <= myhi) {
          double up = (j >= uplo && j <= uphi) ? uprow[j - uplo] : INFINITY;
          double dg = (j - 1 >= uplo && j - 1 <= uphi) ? uprow[j - 1 - uplo] : INFINITY;
          if (i == 0) {
            up = INFINITY;
            dg = (j == 0) ? 0.0 : INFINITY;
          }
          const double lf = (j - 1 >= mylo) ? left : INFINITY;
          const double dt = dchunk[myoff + j - mylo];
          const double cu = __dadd_rn(up, dt), cl = __dadd_rn(lf, dt), cd = __dadd_rn(dg, dt);
          double best = cu;
          unsigned char code = 0;
          if (cl < best) { best = cl; code = 1; }
          if (cd < best) { best = cd; code = 2; }
          dchunk[myoff + j - mylo] = best;
          bp[gofs + j - mylo] = code;
          left = best;
        }
        __syncthreads();
      }
      // hand the last row of the chunk to the next chunk
      const int lr = i0 + R - 1;
      const int lw = (int)hi[lr] - (int)lo[lr] + 1;
      const int lofs = off[lr] - base;
      for (int c = lane; c < lw; c += 64) dprev[c] = dchunk[lofs + c];
      prevlo = (int)lo[lr];
      prevhi = (int)hi[lr];
      __syncthreads();
      i0 += R;
    }
    if (fail) break;
    level_cost = (lty - 1 >= prevlo && lty - 1 <= prevhi) ? dprev[lty - 1 - prevlo] : INFINITY;

    // ---- 3. back-trace (one lane), path written from the end of the buffer ----
    if (lane == 0) {
      int i = ltx - 1, j = lty - 1, pos = pcap;
      int ok = (level_cost < INFINITY) ? 1 : 0;
      int ro = off[i], rl = (int)lo[i];
      while (ok) {
        if (pos == 0) { ok = 0; break; }
        --pos;
        pth_i[pos] = (unsigned short)i;
        pth_j[pos] = (unsigned short)j;
        const unsigned char code = bp[ro + j - rl];
        if (code == 1) {
          j -= 1;
          if (j < rl) ok = 0;
        } else {
          if (code == 2) j -= 1;
          i -= 1;
          if (i < 0) {
            if (j != -1) ok = 0;
            break;
          }
          ro = off[i];
          rl = (int)lo[i];
          if (j < rl || j > (int)hi[i]) ok = 0;
        }
      }
      bcast[0] = pos;
      bcast[1] = ok;
    }
    __syncthreads();
    pstart = bcast[0];
    pn = pcap - pstart;
    if (!bcast[1]) fail = true;
    __syncthreads();
  }

  if (fail) {
    if (lane == 0) {
      p.path_len[n] = 0;
      p.cost[n] = NAN;
    }
    return;
  }
  for (int q = lane; q < pn; q += 64) {
    out_i[q] = pth_i[pstart + q];
    out_j[q] = pth_j[pstart + q];
  }
  if (lane == 0) {
    p.path_len[n] = pn;
    p.cost[n] = level_cost;
  }
}

size_t lds_bytes(int Tx, int Ty, int cellcap, int chunkcap) {
  size_t b = 0;
  b += sizeof(double) * ((size_t)chunkcap + Ty);
  b += sizeof(int) * ((size_t)(Tx + 1) + 2 * kMaxLevels + 4);
  b += sizeof(unsigned short) * ((size_t)2 * Tx + 2 * (Tx / 2 + 2) + 2 * (size_t)(Tx + Ty));
  b += (size_t)cellcap;
  return (b + 15) & ~(size_t)15;
}

}  // namespace

int launch_fastdtw(hipStream_t s, int device, const double *X, const double *Y, const int32_t *lenx,
                   const int32_t *leny, int N, int Tx, int Ty, int D, int radius, int32_t *path_i,
                   int32_t *path_j, int32_t *path_len, double *cost) {
  if (Tx > 65535 || Ty > 65535) {
    set_error("fastdtw: sequences longer than 65535 frames are not supported");
    return MLPG_HIP_EINVAL;
  }
  DtwParams p;
  p.X = X; p.Y = Y; p.lenx = lenx; p.leny = leny;
  p.N = N; p.Tx = Tx; p.Ty = Ty; p.D = D; p.radius = radius;
  p.path_i = path_i; p.path_j = path_j; p.path_len = path_len; p.cost = cost;
  p.pyr_stride = (size_t)(Tx + Ty) * D;
  // window cells per level <= (4r+2)(tx+ty); the coarsest level runs a
  // full DTW with one side <= r+1
  long cc = (long)(4 * radius + 2) * (Tx + Ty) + 64;
  const long full = (long)Tx * Ty;
  if (full < cc) cc = full + 64;
  p.cellcap = (int)cc;
  p.chunkcap = Ty > 1024 ? Ty : 1024;
  const size_t lds = lds_bytes(Tx, Ty, p.cellcap, p.chunkcap);
  if (lds > 160 * 1024) {
    set_error("fastdtw: Tx=%d, Ty=%d, radius=%d needs %zu bytes of LDS (> 160 KiB)", Tx, Ty, radius, lds);
    return MLPG_HIP_EINVAL;
  }
  p.pyr = (double *)scratch(device, 1, sizeof(double) * p.pyr_stride * (size_t)N);
  if (!p.pyr) return MLPG_HIP_ENOMEM;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)fastdtw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  hipLaunchKernelGGL(fastdtw_kernel, dim3(N), dim3(64), lds, s, p);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mlpg
