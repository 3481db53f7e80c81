// Device helpers shared by the generic and the wave-per-system MLPG kernels:
// per-system view of the inputs and the on-the-fly assembly of one column of
// the banded precision matrix (natural order, any window set).
#pragma once
#include "common.h"

namespace mlpg {

template <typename T>
__device__ __forceinline__ double recip_in_dtype(T v);
template <>
__device__ __forceinline__ double recip_in_dtype<float>(float v) {
  return (double)__fdiv_rn(1.0f, v);  // reciprocal evaluated in float32 (_mlpg.py:188)
}
template <>
__device__ __forceinline__ double recip_in_dtype<double>(double v) {
  return 1.0 / v;
}

// One (utterance b, static dim d) system as the kernels see it.
template <typename TIN, bool BWD>
struct SysView {
  const TIN *mean;  // (Tmax, D) rows of utterance b (forward)
  const TIN *var;   // (Tmax, D) rows, or (D,), or nullptr
  const TIN *gout;  // (Tmax, sd) rows (backward)
  int var_mode, sd, d, T, mw;
  long ld_in, ld_gout;  // row strides of mean/var and of gout

  __device__ __forceinline__ double tau(int w, int t) const {
    // zero precision on the edge frames of the dynamic windows; Python's
    // precisions[-0:] slice makes mw == 0 zero the whole column (_mlpg.py:191-193)
    if (w != 0 && (mw == 0 || t < mw || t >= T - mw)) return 0.0;
    if (var_mode == MLPG_HIP_VAR_UNIT) return 1.0;
    const TIN v = (var_mode == MLPG_HIP_VAR_GLOBAL) ? var[w * sd + d] : var[(size_t)t * ld_in + w * sd + d];
    return recip_in_dtype<TIN>(v);
  }
};

template <typename TIN, bool BWD>
__device__ __forceinline__ SysView<TIN, BWD> make_view(const Problem &p, const WinSet &ws, int b, int d, int T) {
  SysView<TIN, BWD> v;
  v.mean = BWD ? nullptr : (const TIN *)p.mean + (size_t)b * p.Tmax * p.ld_in;
  v.var = (const TIN *)p.var;
  if (p.var_mode == MLPG_HIP_VAR_FRAME) v.var += (size_t)b * p.Tmax * p.ld_in;
  v.gout = BWD ? (const TIN *)p.grad_out + (size_t)b * p.Tmax * p.ld_gout : nullptr;
  v.var_mode = p.var_mode;
  v.ld_in = p.ld_in;
  v.ld_gout = p.ld_gout;
  v.sd = p.pitch ? p.pitch : p.sd;  // SysView::sd is the pitch between a dim's windows
  v.d = d;
  v.T = T;
  v.mw = ws.mw;
  return v;
}

// Column f of the lower band: pk[k] = P[f+k, f], k = 0..Q, and the right-hand side.
//   P[f+k, f] = sum_w sum_t c_w[l_w+f-t] c_w[l_w+f+k-t] tau_w[t]
//   rhs[f]    = sum_w sum_t c_w[l_w+f-t] tau_w[t] mu_w[t]   (forward) | grad_out[f, d] (backward)
// Entries that would fall outside the T x T matrix are zero (the "extra
// entries" of the band rectangle, never written by tensor.pyx:82-174).
template <int Q, typename TIN, bool BWD>
__device__ __forceinline__ void assemble_frame(const SysView<TIN, BWD> &v, const WinSet &ws, int f, double (&pk)[Q + 1],
                                               double &rhs) {
#pragma unroll
  for (int k = 0; k <= Q; ++k) pk[k] = 0.0;
  rhs = BWD ? (double)v.gout[(size_t)f * v.ld_gout + v.d] : 0.0;
  const int T = v.T;
  for (int w = 0; w < ws.nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *c = ws.c + ws.off[w];
    const int t0 = f - u < 0 ? 0 : f - u;
    const int t1 = f + l > T - 1 ? T - 1 : f + l;
    for (int t = t0; t <= t1; ++t) {
      const double a = c[l + f - t] * v.tau(w, t);
      if (!BWD) rhs += a * (double)v.mean[(size_t)t * v.ld_in + w * v.sd + v.d];
#pragma unroll
      for (int k = 0; k <= Q; ++k) {
        const int idx = l + f + k - t;
        if (idx <= l + u && f + k < T) pk[k] += a * c[idx];
      }
    }
  }
}

// Natural-order pivot scan: index (1-based) of the first non-positive pivot of
// the banded Cholesky, 0 if none -- the "k-th leading minor" of linalg.pyx:79-82.
template <int Q, typename TIN, bool BWD>
__device__ int first_bad_pivot(const SysView<TIN, BWD> &v, const WinSet &ws) {
  double pend[Q + 1][Q + 1];
#pragma unroll
  for (int j = 0; j <= Q; ++j)
#pragma unroll
    for (int k = 0; k <= Q; ++k) pend[j][k] = 0.0;
  for (int f = 0; f < v.T; ++f) {
    double pk[Q + 1], rhs;
    assemble_frame<Q, TIN, BWD>(v, ws, f, pk, rhs);
    double c[Q + 1];
#pragma unroll
    for (int k = 0; k <= Q; ++k) c[k] = pk[k] + pend[0][k];
    if (c[0] <= 0.0) return f + 1;
    const double iv0 = 1.0 / c[0];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
      for (int l = 0; l <= Q; ++l) {
        double nv = pend[k + 1][l];
        if (l + k + 1 <= Q) nv -= c[l + k + 1] * c[k + 1] * iv0;
        pend[k][l] = nv;
      }
    }
  }
  return 0;
}

}  // namespace mlpg
