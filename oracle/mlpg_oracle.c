/*
 * oracle/mlpg_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, one thread) of the MLPG forward path of the
 * reference r9y9/nnmnkwii.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared object; the product
 * (nnmnkwii_amd) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 * golden vectors produced by the reference itself (tests/golden/make_golden.py
 * imports the unmodified reference built from /root/reference) and against the
 * known-answer vector in SURVEY.md 8(c).
 *
 * Every function cites the reference file:line whose arithmetic and summation
 * order it follows (paths relative to /root/reference/nnmnkwii/).
 *
 * Band storage (paramgen/_bandmat/core.pyx:20-47, full.pyx:26-60): a square
 * matrix with l sub- and u super-diagonals is held as a (l+u+1) x frames
 * row-major rectangle with   full[j + i, j] == rect[u + i, j],  i in [-u, l].
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  FP
 * contraction is disabled so that the summation order below is also the
 * rounding order (the reference's Cython build has no FMA either).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline long lmin(long a, long b) { return a < b ? a : b; }
static inline long lmax(long a, long b) { return a > b ? a : b; }

/* One delta window (l, u, coeff[l+u+1]) as build_win_mats keeps it
 * (paramgen/_mlpg.py:13-50): the window matrix W has W[t, t+k] = coeff[l+k],
 * k in [-l, u]; its band rectangle has constant rows, so the rectangle is never
 * materialised here -- rect[row, *] == coeff[row]. */
typedef struct {
  long l, u;
  const double *coeff;
} win_t;

/* b += W^T . v        -- paramgen/_bandmat/tensor.pyx:20-64 called as
 * dot_mv_plus_equals(win_mat.T, b_frames[:, w], target=b) (_mlpg.py:84).
 * win_mat.T is the non-transposed BandMat(l_a=u_w, u_a=l_w): the loop over
 * o_a in [-u_a, l_a] adds  rect[u_a+o_a, frame-o_a] * v[frame-o_a]. */
static void band_wt_dot_v_acc(const win_t *w, long frames, const double *v,
                              double *target) {
  const long l_a = w->u, u_a = w->l;
  for (long o_a = -u_a; o_a <= l_a; ++o_a) {
    const double c = w->coeff[u_a + o_a];
    const long lo = lmax(0, o_a), hi = lmax(0, frames + lmin(0, o_a));
    for (long f = lo; f < hi; ++f) target[f] += c * v[f - o_a];
  }
}

/* P += band_{sdw}( W^T diag(tau) W )   -- tensor.pyx:82-174 called as
 * dot_mm_plus_equals(win_mat.T, win_mat, target_bm=prec, diag=tau) (_mlpg.py:85-87).
 * a = W^T (plain, l_a=u_w, u_a=l_w), b = W (transposed flag set, l_b=l_w,
 * u_b=u_w), c = prec (plain, l_c=u_c=sdw).  prec is (2*sdw+1) x frames. */
static void band_wt_diag_w_acc(const win_t *w, long frames, const double *tau,
                               long sdw, double *prec) {
  const long l_a = w->u, u_a = w->l, l_b = w->l, u_b = w->u;
  const long l_c = sdw, u_c = sdw;
  for (long o_c = -lmin(u_c, u_a + u_b); o_c <= lmin(l_c, l_a + l_b); ++o_c) {
    for (long o_a = -lmin(u_a, l_b - o_c); o_a <= lmin(l_a, u_b + o_c); ++o_a) {
      const long o_b = o_c - o_a;
      const double ca = w->coeff[u_a + o_a]; /* row_a = u_a + o_a           */
      const double cb = w->coeff[l_b - o_b]; /* row_b = l_b - o_b (b is .T) */
      double *crow = prec + (u_c + o_c) * frames;
      const long lo = lmax(0, lmax(-o_a, o_b));
      const long hi = lmax(0, frames + lmin(0, lmin(-o_a, o_b)));
      for (long f = lo; f < hi; ++f) crow[f - o_b] += ca * cb * tau[f];
    }
  }
}

/* In-place lower banded Cholesky of a (depth+1) x frames rectangle holding the
 * diagonal (row 0) and sub-diagonals  -- linalg.pyx:36-104 with lower=True.
 * Returns 0, or frame+1 of the first non-positive pivot (the reference raises
 * LinAlgError('%d-th leading minor not positive definite'), linalg.pyx:79-82). */
static long cholesky_banded_lower(double *mat, long depth, long frames) {
  double v[64];
  for (long f = 0; f < frames; ++f) {
    const double v0 = mat[f];
    if (v0 <= 0.0) return f + 1;
    const double iv0 = 1.0 / v0;
    const double siv0 = sqrt(iv0);
    for (long k = 0; k < depth; ++k) v[k] = mat[(k + 1) * frames + f];
    mat[f] = 1.0 / siv0;
    for (long k = 0; k < depth; ++k) mat[(k + 1) * frames + f] = v[k] * siv0;
    const long kmax = lmin(depth, frames - f - 1);
    for (long k = 0; k < kmax; ++k)
      for (long l = 0; l < depth - k; ++l)
        mat[l * frames + (k + f + 1)] -= v[l + k] * v[k] * iv0;
  }
  return 0;
}

/* Triangular banded solves with the lower factor L -- linalg.pyx:106-176.
 * forward:  L x = b    (lower=True, transposed=False: branch :150-154)
 * backward: L^T x = b  (lower=True, transposed=True:  branch :145-149)
 * Returns 0, or frame+1 where the diagonal is exactly zero (:170-173). */
static long solve_lower_forward(const double *chol, long depth, long frames,
                                const double *b, double *x) {
  for (long pos = 0; pos < frames; ++pos) {
    const long f = pos;
    double diff = b[f];
    const long kmax = lmin(depth + 1, pos + 1);
    for (long k = 1; k < kmax; ++k) diff -= chol[k * frames + (f - k)] * x[f - k];
    const double denom = chol[f];
    if (denom == 0.0) return f + 1;
    x[f] = diff / denom;
  }
  return 0;
}

static long solve_lower_transposed(const double *chol, long depth, long frames,
                                   const double *b, double *x) {
  for (long pos = 0; pos < frames; ++pos) {
    const long f = frames - 1 - pos;
    double diff = b[f];
    const long kmax = lmin(depth + 1, pos + 1);
    for (long k = 1; k < kmax; ++k) diff -= chol[k * frames + f] * x[f + k];
    const double denom = chol[f];
    if (denom == 0.0) return f + 1;
    x[f] = diff / denom;
  }
  return 0;
}

/* Shared driver for paramgen.mlpg (_mlpg.py:92-199) on one utterance.
 *   mean, var : (T, D) row-major, D = num_windows * static_dim, column w*sd+d
 *   in_is_f32 : inputs are float32 -> the reciprocal 1/var is evaluated in
 *               float32 and only then widened (:188 assigns a float32 array
 *               into the float64 workspace); everything after is float64.
 *   out       : (T, static_dim) in the input dtype (:166,183,197).
 * status (may be NULL): per static dim, 0 or k = "k-th leading minor not PD";
 * the function stops at the first failing dim like the reference's raise and
 * returns that dim's index + 1 (0 on success).
 */
static long mlpg_one(const void *mean, const void *var, int in_is_f32, long T,
                     long D, long num_windows, const win_t *wins, void *out,
                     int32_t *status) {
  const long sd = D / num_windows;
  long sdw = 0, mw = 0;
  for (long w = 0; w < num_windows; ++w) {
    sdw = lmax(sdw, wins[w].l + wins[w].u);            /* _mlpg.py:72-73 */
    mw = lmax(mw, lmax(wins[w].l, wins[w].u));         /* :177           */
  }
  double *mu = (double *)malloc(sizeof(double) * T * num_windows);
  double *tau = (double *)malloc(sizeof(double) * T * num_windows);
  double *bs = (double *)malloc(sizeof(double) * T);
  double *b = (double *)malloc(sizeof(double) * T);
  double *prec = (double *)malloc(sizeof(double) * T * (2 * sdw + 1));
  double *z = (double *)malloc(sizeof(double) * T);
  double *x = (double *)malloc(sizeof(double) * T);
  long rc = 0;
  if (status) memset(status, 0, sizeof(int32_t) * sd);

  for (long d = 0; d < sd && rc == 0; ++d) {
    /* column gather + reciprocal + edge zeroing, :186-193 (workspace is
     * window-major here; the reference's (T, num_windows) layout only changes
     * strides, not values) */
    for (long w = 0; w < num_windows; ++w) {
      const long col = w * sd + d;
      for (long t = 0; t < T; ++t) {
        double m, p;
        if (in_is_f32) {
          m = (double)((const float *)mean)[t * D + col];
          p = (double)(1.0f / ((const float *)var)[t * D + col]);
        } else {
          m = ((const double *)mean)[t * D + col];
          p = 1.0 / ((const double *)var)[t * D + col];
        }
        /* precisions[:mw] = 0; precisions[-mw:] = 0 for w != 0.  Python slice
         * semantics: [-0:] is the whole column (SURVEY 8c "-0: quirk"). */
        if (w != 0 && (mw == 0 || t < mw || t >= T - mw)) p = 0.0;
        mu[w * T + t] = m;
        tau[w * T + t] = p;
      }
    }
    /* build_poe, :53-89 */
    memset(b, 0, sizeof(double) * T);
    memset(prec, 0, sizeof(double) * T * (2 * sdw + 1));
    for (long w = 0; w < num_windows; ++w) {
      for (long t = 0; t < T; ++t) bs[t] = tau[w * T + t] * mu[w * T + t]; /* :195 */
      band_wt_dot_v_acc(&wins[w], T, bs, b);
      band_wt_diag_w_acc(&wins[w], T, tau + w * T, sdw, prec);
    }
    /* solveh, linalg.pyx:290-304: lower half-band rows sdw..2*sdw (:216-219) */
    double *half = prec + sdw * T;
    long bad = cholesky_banded_lower(half, sdw, T);
    if (bad) {
      if (status) status[d] = (int32_t)bad;
      rc = d + 1;
      break;
    }
    solve_lower_forward(half, sdw, T, b, z);
    solve_lower_transposed(half, sdw, T, z, x);
    for (long t = 0; t < T; ++t) {
      if (in_is_f32)
        ((float *)out)[t * sd + d] = (float)x[t];
      else
        ((double *)out)[t * sd + d] = x[t];
    }
  }
  free(mu); free(tau); free(bs); free(b); free(prec); free(z); free(x);
  return rc;
}

static win_t *unpack_windows(long num_windows, const int32_t *win_l,
                             const int32_t *win_u, const double *win_coef) {
  win_t *w = (win_t *)malloc(sizeof(win_t) * num_windows);
  long off = 0;
  for (long i = 0; i < num_windows; ++i) {
    w[i].l = win_l[i];
    w[i].u = win_u[i];
    w[i].coeff = win_coef + off;
    off += win_l[i] + win_u[i] + 1;
  }
  return w;
}

/* Batched entry points: the reference batches by a Python loop over the
 * utterances of a zero-padded (B, Tmax, D) array (util/__init__.py:44-66);
 * lengths == NULL means every utterance has Tmax frames.  var_is_global: var is
 * (D,) and is tiled over frames (_mlpg.py:169-170).  out is (B, Tmax, sd);
 * frames >= lengths[b] are zero-filled.  status is (B, sd).  Returns 0 or
 * 1 + index of the first failing (b*sd + d). */
static long mlpg_batch(const void *mean, const void *var, int in_is_f32,
                       int var_is_global, const int32_t *lengths, long B,
                       long Tmax, long D, long num_windows,
                       const int32_t *win_l, const int32_t *win_u,
                       const double *win_coef, void *out, int32_t *status) {
  win_t *wins = unpack_windows(num_windows, win_l, win_u, win_coef);
  const long sd = D / num_windows;
  const size_t esz = in_is_f32 ? 4 : 8;
  long rc = 0;
  char *vtile = NULL;
  if (var_is_global) vtile = (char *)malloc(esz * Tmax * D);
  for (long b = 0; b < B; ++b) {
    const long T = lengths ? lengths[b] : Tmax;
    const char *m = (const char *)mean + esz * b * Tmax * D;
    const char *v;
    if (var_is_global) {
      for (long t = 0; t < T; ++t) memcpy(vtile + esz * t * D, var, esz * D);
      v = vtile;
    } else {
      v = (const char *)var + esz * b * Tmax * D;
    }
    char *o = (char *)out + esz * b * Tmax * sd;
    memset(o, 0, esz * Tmax * sd);
    long r = mlpg_one(m, v, in_is_f32, T, D, num_windows, wins, o,
                      status ? status + b * sd : NULL);
    if (r && !rc) rc = b * sd + r;
  }
  free(vtile);
  free(wins);
  return rc;
}

ORACLE_API long oracle_mlpg_f64(const double *mean, const double *var,
                                int var_is_global, const int32_t *lengths,
                                long B, long Tmax, long D, long num_windows,
                                const int32_t *win_l, const int32_t *win_u,
                                const double *win_coef, double *out,
                                int32_t *status) {
  return mlpg_batch(mean, var, 0, var_is_global, lengths, B, Tmax, D,
                    num_windows, win_l, win_u, win_coef, out, status);
}

ORACLE_API long oracle_mlpg_f32(const float *mean, const float *var,
                                int var_is_global, const int32_t *lengths,
                                long B, long Tmax, long D, long num_windows,
                                const int32_t *win_l, const int32_t *win_u,
                                const double *win_coef, float *out,
                                int32_t *status) {
  return mlpg_batch(mean, var, 1, var_is_global, lengths, B, Tmax, D,
                    num_windows, win_l, win_u, win_coef, out, status);
}

/* Intermediate products for one (utterance, static dim): b, the full
 * (2*sdw+1) x T precision band and the lower Cholesky rectangle -- lets the
 * tests pin the restatement stage by stage against SURVEY 8(c)'s vector. */
ORACLE_API long oracle_mlpg_stages_f64(const double *mean, const double *var,
                                       long T, long D, long d,
                                       long num_windows, const int32_t *win_l,
                                       const int32_t *win_u,
                                       const double *win_coef, double *b_out,
                                       double *prec_out, double *chol_out) {
  win_t *wins = unpack_windows(num_windows, win_l, win_u, win_coef);
  const long sd = D / num_windows;
  long sdw = 0, mw = 0;
  for (long w = 0; w < num_windows; ++w) {
    sdw = lmax(sdw, wins[w].l + wins[w].u);
    mw = lmax(mw, lmax(wins[w].l, wins[w].u));
  }
  double *tau = (double *)malloc(sizeof(double) * T);
  double *bs = (double *)malloc(sizeof(double) * T);
  memset(b_out, 0, sizeof(double) * T);
  memset(prec_out, 0, sizeof(double) * T * (2 * sdw + 1));
  for (long w = 0; w < num_windows; ++w) {
    const long col = w * sd + d;
    for (long t = 0; t < T; ++t) {
      double p = 1.0 / var[t * D + col];
      if (w != 0 && (mw == 0 || t < mw || t >= T - mw)) p = 0.0;
      tau[t] = p;
      bs[t] = p * mean[t * D + col];
    }
    band_wt_dot_v_acc(&wins[w], T, bs, b_out);
    band_wt_diag_w_acc(&wins[w], T, tau, sdw, prec_out);
  }
  memcpy(chol_out, prec_out + sdw * T, sizeof(double) * T * (sdw + 1));
  long bad = cholesky_banded_lower(chol_out, sdw, T);
  free(tau); free(bs); free(wins);
  return bad;
}
