// fastdtw from HOST-evaluated local costs: the DP recurrence, the back-trace and the window expansion of one resolution
// level on the GPU, for `dist` callables of DTWAligner that no device-side cost reproduces
// (reference: preprocessing/alignment.py:35-38 accepts any Python callable and hands it to fastdtw, :50).
//
// The host (nnmnkwii_amd/preprocessing/alignment.py) walks the levels from the coarsest to the finest; per level
//   1. dtw_window_kernel  per-row windows [lo_i, hi_i] of every pair from its path at the coarser level (interval form
//                         of upstream's __expand_window, as csrc/dtw_fast.hip) and the offset of every row's first cell
//                         in the pair's cost buffer; a pair whose recursion bottoms out at this level gets the full
//                         window, a pair that has not started yet (level_len 0) is skipped;
//   2. the host evaluates dist(x_i, y_j) for the window's cells with the user's callable and uploads them;
//   3. dtw_costs_kernel   D[i,j] = cost + the tie rule's pick of (D[i-1,j], D[i,j-1], D[i-1,j-1]) -- compared after the
//                         add, as upstream's __dtw --, back-pointers, back-trace; the path stays on the device for the
//                         next level's windows.
// One lane per pair: the host-side cost evaluation (one interpreter call per cell) outweighs this by orders of magnitude.
#include <math.h>

#include "common.h"

namespace mlpg {
namespace {

// rows: per pair `row_stride` entries of (lo, hi); offs: per pair row_stride + 1 prefix sums of the widths.
__global__ __launch_bounds__(64) void dtw_window_kernel(int N, int radius, const int32_t *__restrict__ ltx,
                                                        const int32_t *__restrict__ lty, const int32_t *__restrict__ full,
                                                        const int32_t *__restrict__ cpath_i,
                                                        const int32_t *__restrict__ cpath_j,
                                                        const int32_t *__restrict__ cpath_len, int cpath_stride,
                                                        int32_t *__restrict__ row_lo, int32_t *__restrict__ row_hi,
                                                        int64_t *__restrict__ row_off, int row_stride,
                                                        int32_t *__restrict__ cfirst, int32_t *__restrict__ clast) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const int tx = ltx[n], ty = lty[n];
  int32_t *lo = row_lo + (size_t)n * row_stride, *hi = row_hi + (size_t)n * row_stride;
  int64_t *off = row_off + (size_t)n * (row_stride + 1);
  if (tx <= 0) {
    off[0] = 0;
    return;
  }
  if (full[n]) {
    for (int i = 0; i < tx; ++i) { lo[i] = 0; hi[i] = ty - 1; }
  } else {
    // first / last column of the coarse path in every coarse row (the path is monotone and visits every row)
    const int cx = tx >> 1, r = radius;
    int32_t *cf = cfirst + (size_t)n * row_stride, *cl = clast + (size_t)n * row_stride;
    const int32_t *pi = cpath_i + (size_t)n * cpath_stride, *pj = cpath_j + (size_t)n * cpath_stride;
    const int pn = cpath_len[n];
    for (int q = 0; q < pn; ++q) {
      if (q == 0 || pi[q - 1] != pi[q]) cf[pi[q]] = pj[q];
      if (q == pn - 1 || pi[q + 1] != pi[q]) cl[pi[q]] = pj[q];
    }
    for (int i = 0; i < tx; ++i) {
      const int ci = i >> 1;
      const int r0 = ci - r < 0 ? 0 : ci - r;
      const int r1 = ci + r > cx - 1 ? cx - 1 : ci + r;
      const int a = 2 * (cf[r0] - r), b = 2 * (cl[r1] + r) + 1;
      lo[i] = a < 0 ? 0 : a;
      hi[i] = b > ty - 1 ? ty - 1 : b;
    }
  }
  int64_t run = 0;
  for (int i = 0; i < tx; ++i) {
    off[i] = run;
    run += hi[i] - lo[i] + 1;
  }
  off[tx] = run;
}

template <int TIE>
__global__ __launch_bounds__(64) void dtw_costs_kernel(int N, const int32_t *__restrict__ ltx, const int32_t *__restrict__ lty,
                                                       const int32_t *__restrict__ row_lo, const int32_t *__restrict__ row_hi,
                                                       const int64_t *__restrict__ row_off, int row_stride,
                                                       const double *__restrict__ costs, const int64_t *__restrict__ cost_base,
                                                       double *__restrict__ drow, int dstride, unsigned char *__restrict__ bp,
                                                       const int64_t *__restrict__ bp_base, int32_t *__restrict__ path_i,
                                                       int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                                                       int path_stride, double *__restrict__ cost_out) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const int tx = ltx[n], ty = lty[n];
  if (tx <= 0) return;  // this pair's recursion has not reached the level yet
  const int32_t *lo = row_lo + (size_t)n * row_stride, *hi = row_hi + (size_t)n * row_stride;
  const int64_t *off = row_off + (size_t)n * (row_stride + 1);
  const double *c = costs + cost_base[n];
  unsigned char *b = bp + bp_base[n];
  double *d0 = drow + (size_t)n * 2 * dstride, *d1 = d0 + dstride;  // the row above / the current row, by column
  const double INF = INFINITY;
  int plo = 0, phi = -1;  // the window of the row above (row -1: only the virtual cell (-1, -1) = 0)
  for (int i = 0; i < tx; ++i) {
    const int l = lo[i], h = hi[i];
    double left = INF;
    for (int j = l; j <= h; ++j) {
      const double dt = c[off[i] + (j - l)];
      double up, dg;
      if (i == 0) {
        up = INF;
        dg = j == 0 ? 0.0 : INF;
      } else {
        up = (j >= plo && j <= phi) ? d0[j] : INF;
        dg = (j - 1 >= plo && j - 1 <= phi) ? d0[j - 1] : INF;
      }
      const double cu = up + dt, cl = left + dt, cd = dg + dt;
      double best;
      unsigned char p;
      if (TIE == MLPG_HIP_TIE_FIRST_MIN) {
        best = cu; p = 0;
        if (cl < best) { best = cl; p = 1; }
        if (cd < best) { best = cd; p = 2; }
      } else {
        if (cu < cl && cu < cd) { best = cu; p = 0; }
        else if (cl < cd) { best = cl; p = 1; }
        else { best = cd; p = 2; }
      }
      d1[j] = best;
      b[off[i] + (j - l)] = p;
      left = best;
    }
    double *t = d0; d0 = d1; d1 = t;
    plo = l; phi = h;
  }
  // back-trace from the corner
  int32_t *pi = path_i + (size_t)n * path_stride, *pj = path_j + (size_t)n * path_stride;
  int i = tx - 1, j = ty - 1, len = 0;
  const bool reach = j >= lo[i] && j <= hi[i];
  const double total = reach ? d0[j] : INF;
  bool ok = reach && total < INF;
  while (ok && !(i == -1 && j == -1)) {
    if (i < 0 || j < 0 || j < lo[i] || j > hi[i] || len >= path_stride) { ok = false; break; }
    pi[len] = i;
    pj[len] = j;
    ++len;
    const unsigned char p = b[off[i] + (j - lo[i])];
    if (p == 0) i -= 1;
    else if (p == 1) j -= 1;
    else { i -= 1; j -= 1; }
  }
  if (!ok) {
    path_len[n] = 0;
    cost_out[n] = INF;
    return;
  }
  for (int q = 0; q < len / 2; ++q) {
    const int32_t a = pi[q], e = pj[q];
    pi[q] = pi[len - 1 - q]; pj[q] = pj[len - 1 - q];
    pi[len - 1 - q] = a; pj[len - 1 - q] = e;
  }
  path_len[n] = len;
  cost_out[n] = total;
}

}  // namespace

int launch_dtw_window(hipStream_t s, int device, int N, int radius, const int32_t *ltx, const int32_t *lty,
                      const int32_t *full, const int32_t *cpath_i, const int32_t *cpath_j, const int32_t *cpath_len,
                      int cpath_stride, int32_t *row_lo, int32_t *row_hi, int64_t *row_off, int row_stride) {
  int32_t *tmp = (int32_t *)scratch(device, s, 5, sizeof(int32_t) * 2 * (size_t)N * row_stride);
  if (!tmp) return MLPG_HIP_ENOMEM;
  hipLaunchKernelGGL(dtw_window_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, N, radius, ltx, lty, full, cpath_i,
                     cpath_j, cpath_len, cpath_stride, row_lo, row_hi, row_off, row_stride, tmp, tmp + (size_t)N * row_stride);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_dtw_costs(hipStream_t s, int device, int N, int tie_rule, const int32_t *ltx, const int32_t *lty,
                     const int32_t *row_lo, const int32_t *row_hi, const int64_t *row_off, int row_stride, int max_ty,
                     const double *costs, const int64_t *cost_base, int64_t total_cells, int32_t *path_i, int32_t *path_j,
                     int32_t *path_len, int path_stride, double *cost_out) {
  // scratch: two D rows per pair, one back-pointer byte per window cell (same bases as the costs)
  const size_t dbytes = sizeof(double) * 2 * (size_t)N * max_ty;
  char *sc = (char *)scratch(device, s, 5, dbytes + (size_t)total_cells + 64);
  if (!sc) return MLPG_HIP_ENOMEM;
  double *drow = (double *)sc;
  unsigned char *bp = (unsigned char *)(sc + dbytes);
  if (tie_rule == MLPG_HIP_TIE_FIRST_MIN)
    hipLaunchKernelGGL((dtw_costs_kernel<MLPG_HIP_TIE_FIRST_MIN>), dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, N, ltx, lty,
                       row_lo, row_hi, row_off, row_stride, costs, cost_base, drow, max_ty, bp, cost_base, path_i, path_j,
                       path_len, path_stride, cost_out);
  else
    hipLaunchKernelGGL((dtw_costs_kernel<MLPG_HIP_TIE_DIAG_LAST>), dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, N, ltx, lty,
                       row_lo, row_hi, row_off, row_stride, costs, cost_base, drow, max_ty, bp, cost_base, path_i, path_j,
                       path_len, path_stride, cost_out);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mlpg
