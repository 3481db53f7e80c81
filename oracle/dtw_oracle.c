/*
 * oracle/dtw_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * CPU restatement of fastdtw(x, y, radius, dist=L2) as the reference's
 * DTWAligner calls it (/root/reference/nnmnkwii/preprocessing/alignment.py:50,
 * default dist = lambda x, y: norm(x - y), :35).
 *
 * PARITY UNPINNED.  The arithmetic lives in the third-party PyPI package
 * `fastdtw` (github.com/slaypni/fastdtw), which the reference depends on
 * without a version pin (setup.py:139 is the bare string "fastdtw"), which is
 * not vendored under /root/reference and is not installed in this image.  The
 * reference's own tests pin only output shapes and a norm inequality
 * (tests/test_preprocessing.py:441-501).  This file therefore restates the
 * published algorithm (Salvador & Chan, "FastDTW: Toward accurate dynamic time
 * warping in linear time and space", 2007) with the semantics of upstream's
 * pure-Python fastdtw/fastdtw.py as recalled in SURVEY.md 8(c):
 *   - __fastdtw: min_time_size = radius + 2; below it, full DTW; else halve both
 *     series (pairwise mean, odd tail dropped), recurse, expand the coarse path
 *     by `radius`, run windowed DTW.
 *   - __expand_window: union of (2r+1)^2 neighbourhoods of path cells, each
 *     coarse cell -> its 2x2 fine cells, then per fine row scan j from the
 *     previous row's first column and keep the first contiguous run.
 *   - __dtw: row-major over the window; D[0,0] = 0, absent cells = +inf;
 *     D[i,j] = min(D[i-1,j]+dt, D[i,j-1]+dt, D[i-1,j-1]+dt), FIRST minimum wins
 *     (Python min(key=)), compared AFTER adding dt; back-trace from the corner.
 * Tie rule chosen here: up (i-1,j), then left (i,j-1), then diagonal.
 *
 * Local cost: sqrt(sum_k (x_k - y_k)^2), float64, k ascending, separate
 * multiply and add (no FMA: build with -ffp-contract=off) -- the HIP kernel
 * uses the same order so costs compare bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static double l2_cost(const double *a, const double *b, long D) {
  double acc = 0.0;
  for (long k = 0; k < D; ++k) {
    const double diff = a[k] - b[k];
    acc = acc + diff * diff;
  }
  return sqrt(acc);
}

/* Windowed DTW over per-row column intervals [lo[i], hi[i]] (0-based, inclusive).
 * Returns path length; path written front-to-back.  pred codes: 0 up, 1 left,
 * 2 diag. */
/* tie = 0: upstream's pure-Python __dtw (FIRST minimum of up, left, diagonal).
 * tie = 1: the strict-less chain recalled for upstream's compiled _fastdtw -- up
 * only if it beats both others, else left only if it beats the diagonal, else
 * the diagonal.  UNVERIFIED: the package cannot be installed
 * here; tests/test_dtw_gpu.py reports which rule an installed fastdtw follows. */
static long g_tie = 0;
static long dtw_windowed(const double *x, long tx, const double *y, long ty,
                         long D, const long *lo, const long *hi,
                         int32_t *path_i, int32_t *path_j, double *cost_out) {
  long *off = (long *)malloc(sizeof(long) * (tx + 1));
  long total = 0;
  for (long i = 0; i < tx; ++i) {
    off[i] = total;
    total += hi[i] >= lo[i] ? hi[i] - lo[i] + 1 : 0;
  }
  off[tx] = total;
  double *cost = (double *)malloc(sizeof(double) * (total ? total : 1));
  uint8_t *pred = (uint8_t *)malloc(total ? total : 1);
  const double INF = INFINITY;
#define CELL(i, j) \
  (((i) < 0 || (j) < 0) ? (((i) == -1 && (j) == -1) ? 0.0 : INF) \
   : (((j) < lo[i] || (j) > hi[i]) ? INF : cost[off[i] + (j) - lo[i]]))
  for (long i = 0; i < tx; ++i) {
    for (long j = lo[i]; j <= hi[i]; ++j) {
      const double dt = l2_cost(x + i * D, y + j * D, D);
      const double up = CELL(i - 1, j) + dt;
      const double left = CELL(i, j - 1) + dt;
      const double diag = CELL(i - 1, j - 1) + dt;
      double best = up;
      uint8_t p = 0;
      if (g_tie == 0) {
        if (left < best) { best = left; p = 1; }
        if (diag < best) { best = diag; p = 2; }
      } else if (!(up < left && up < diag)) {
        if (left < diag) { best = left; p = 1; }
        else { best = diag; p = 2; }
      }
      cost[off[i] + j - lo[i]] = best;
      pred[off[i] + j - lo[i]] = p;
    }
  }
  long n = 0;
  long i = tx - 1, j = ty - 1;
  *cost_out = CELL(i, j);
  /* back-trace into the tail of the buffers, then shift to the front */
  const long cap = tx + ty;
  int32_t *ri = (int32_t *)malloc(sizeof(int32_t) * cap);
  int32_t *rj = (int32_t *)malloc(sizeof(int32_t) * cap);
  while (!(i == -1 && j == -1)) {
    if (i < 0 || j < 0 || j < lo[i] || j > hi[i] || n >= cap) { n = -1; break; }
    ri[n] = (int32_t)i;
    rj[n] = (int32_t)j;
    ++n;
    const uint8_t p = pred[off[i] + j - lo[i]];
    if (p == 0) { i -= 1; }
    else if (p == 1) { j -= 1; }
    else { i -= 1; j -= 1; }
  }
#undef CELL
  for (long k = 0; k < n; ++k) {
    path_i[k] = ri[n - 1 - k];
    path_j[k] = rj[n - 1 - k];
  }
  free(ri); free(rj); free(off); free(cost); free(pred);
  return n;
}

static long fastdtw_rec(const double *x, long tx, const double *y, long ty,
                        long D, long radius, int32_t *path_i, int32_t *path_j,
                        double *cost_out) {
  const long min_time_size = radius + 2;
  long *lo = (long *)malloc(sizeof(long) * (tx > 0 ? tx : 1));
  long *hi = (long *)malloc(sizeof(long) * (tx > 0 ? tx : 1));
  long n;
  if (tx < min_time_size || ty < min_time_size) {
    for (long i = 0; i < tx; ++i) { lo[i] = 0; hi[i] = ty - 1; }
    n = dtw_windowed(x, tx, y, ty, D, lo, hi, path_i, path_j, cost_out);
    free(lo); free(hi);
    return n;
  }
  /* __reduce_by_half */
  const long cx = tx / 2, cy = ty / 2;
  double *xs = (double *)malloc(sizeof(double) * cx * D);
  double *ys = (double *)malloc(sizeof(double) * cy * D);
  for (long i = 0; i < cx; ++i)
    for (long k = 0; k < D; ++k) xs[i * D + k] = (x[(2 * i) * D + k] + x[(2 * i + 1) * D + k]) / 2;
  for (long i = 0; i < cy; ++i)
    for (long k = 0; k < D; ++k) ys[i * D + k] = (y[(2 * i) * D + k] + y[(2 * i + 1) * D + k]) / 2;
  int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * (cx + cy + 1));
  int32_t *cj = (int32_t *)malloc(sizeof(int32_t) * (cx + cy + 1));
  double ccost;
  const long cn = fastdtw_rec(xs, cx, ys, cy, D, radius, ci, cj, &ccost);
  free(xs); free(ys);
  if (cn < 0) { free(ci); free(cj); free(lo); free(hi); return -1; }

  /* __expand_window, literal: a membership grid over coarse cells shifted by
   * +radius so that negative neighbours are representable. */
  const long gx = cx + 2 * radius + 1, gy = cy + 2 * radius + 1;
  uint8_t *grid = (uint8_t *)calloc((size_t)gx * gy, 1);
  for (long k = 0; k < cn; ++k)
    for (long a = -radius; a <= radius; ++a)
      for (long b = -radius; b <= radius; ++b)
        grid[(ci[k] + a + radius) * gy + (cj[k] + b + radius)] = 1;
  free(ci); free(cj);
#define IN_WINDOW(i, j) \
  (((i) / 2 + radius) < gx && ((j) / 2 + radius) < gy && grid[((i) / 2 + radius) * gy + ((j) / 2 + radius)])
  long start_j = 0;
  int ok = 1;
  for (long i = 0; i < tx; ++i) {
    long new_start = -1, last = -1;
    for (long j = start_j; j < ty; ++j) {
      if (IN_WINDOW(i, j)) {
        if (new_start < 0) new_start = j;
        last = j;
      } else if (new_start >= 0) {
        break;
      }
    }
    if (new_start < 0) { ok = 0; break; } /* upstream would raise TypeError here */
    lo[i] = new_start;
    hi[i] = last;
    start_j = new_start;
  }
#undef IN_WINDOW
  free(grid);
  if (!ok) { free(lo); free(hi); return -1; }
  n = dtw_windowed(x, tx, y, ty, D, lo, hi, path_i, path_j, cost_out);
  free(lo); free(hi);
  return n;
}

/* x: (tx, D), y: (ty, D) row-major float64.  path_i/path_j need tx+ty entries.
 * Returns path length (>= max(tx,ty), <= tx+ty-1) or -1. */
ORACLE_API long oracle_fastdtw_l2(const double *x, long tx, const double *y,
                                  long ty, long D, long radius,
                                  int32_t *path_i, int32_t *path_j,
                                  double *cost) {
  if (tx <= 0 || ty <= 0) return -1;
  return fastdtw_rec(x, tx, y, ty, D, radius, path_i, path_j, cost);
}

/* the same with the tie rule selected (0 / 1, see dtw_windowed); not re-entrant */
ORACLE_API long oracle_fastdtw_l2_tie(const double *x, long tx, const double *y,
                                      long ty, long D, long radius, long tie,
                                      int32_t *path_i, int32_t *path_j,
                                      double *cost) {
  if (tx <= 0 || ty <= 0) return -1;
  g_tie = tie;
  const long n = fastdtw_rec(x, tx, y, ty, D, radius, path_i, path_j, cost);
  g_tie = 0;
  return n;
}

/* The window (per-row [lo, hi]) fastdtw would use at the FINEST level -- lets
 * tests compare the HIP kernel's interval form of __expand_window directly. */
ORACLE_API long oracle_full_dtw_l2(const double *x, long tx, const double *y,
                                   long ty, long D, int32_t *path_i,
                                   int32_t *path_j, double *cost) {
  long *lo = (long *)malloc(sizeof(long) * tx);
  long *hi = (long *)malloc(sizeof(long) * tx);
  for (long i = 0; i < tx; ++i) { lo[i] = 0; hi[i] = ty - 1; }
  long n = dtw_windowed(x, tx, y, ty, D, lo, hi, path_i, path_j, cost);
  free(lo); free(hi);
  return n;
}
