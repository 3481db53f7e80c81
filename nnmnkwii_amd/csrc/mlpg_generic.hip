// Generic MLPG kernels: one THREAD per (utterance, static dim) system, any
// window set with half-bandwidth <= 8, any T.  Adjacent lanes take adjacent
// static dims so that every global access is a contiguous run of lanes
// (feature columns are window-major: column w*sd + d).
//
// This is the fallback / cross-check path (algo = MLPG_HIP_ALGO_GENERIC); the
// wave-per-system kernel in mlpg_wave.hip is the fast path for the common case.
// The banded factor does not fit on chip at one system per lane, so it goes to
// an HBM scratch laid out [frame][row][system] (system fastest = coalesced).
// Three stages: the assembly of P and b is embarrassingly parallel (one thread
// per (frame, system)); only the factorisation + substitutions are sequential
// in time (one thread per system, scratch rows prefetched four frames ahead so
// that the recurrence never waits for memory); the backward's epilogue is
// parallel again.
//
// Math (reference: paramgen/_mlpg.py:92-199, _bandmat/linalg.pyx:36-176):
//   P[f+k, f] = sum_w sum_t c_w[l_w+f-t] c_w[l_w+f+k-t] tau_w[t]
//   rhs[f]    = sum_w sum_t c_w[l_w+f-t] tau_w[t] mu_w[t]      (forward)
//             = grad_out[f, d]                                  (backward)
//   right-looking banded Cholesky fused with the forward substitution, then a
//   reverse sweep for L^T x = z.  Backward epilogue:
//   grad[t, w*sd+d] = tau_w[t] * sum_k c_w[l_w+k] x[t+k]   (paramgen/_mlpg.py:202-281)
#include "assemble.h"

namespace mlpg {
namespace {

// ---- stage 1: assembly, fully parallel: one thread per (frame, system) ----
// scratch[(f * R + k) * S + s], R = Q + 2:  k = 0..Q -> P[f+k, f],  k = Q+1 -> right-hand side of frame f
template <int Q, typename TIN, bool BWD>
__global__ __launch_bounds__(256) void generic_assemble_kernel(Problem p, WinSet ws, double *__restrict__ scratch, long S) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= S * p.Tmax) return;
  const long s = e % S;
  const int f = (int)(e / S);
  const int sd = p.sd;
  const int b = (int)(s / sd), d = (int)(s % sd);
  int T = p.lengths ? p.lengths[b] : p.Tmax;
  T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
  if (f >= T) return;
  const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
  double pk[Q + 1], rhs;
  assemble_frame<Q, TIN, BWD>(view, ws, f, pk, rhs);
  constexpr int R = Q + 2;
  double *sc = scratch + ((size_t)f * R) * S + s;
#pragma unroll
  for (int k = 0; k <= Q; ++k) sc[(size_t)k * S] = pk[k];
  sc[(size_t)(Q + 1) * S] = rhs;
}

// ---- stage 2: the sequential part, one thread per system: right-looking banded Cholesky fused with the
// forward substitution, then the reverse sweep.  The rows of the next kPF frames are loaded while the current
// ones are processed (the recurrence itself never waits for memory).  Overwrites the scratch in place:
// k = 0 -> 1 / L_ff, k = 1..Q -> L[f+k, f], k = Q+1 -> z_f, and after the reverse sweep x_f (backward only).
template <int Q>
constexpr int prefetch_depth() { return Q <= 2 ? 8 : 4; }  // frames in flight per thread (register budget)

template <int Q, typename TIN, typename TOUT, bool BWD>
__global__ __launch_bounds__(64) void generic_kernel(Problem p, WinSet ws, double *__restrict__ scratch, long S) {
  const long s = (long)blockIdx.x * 64 + threadIdx.x;
  if (s >= S) return;
  const int sd = p.sd, Tmax = p.Tmax;
  const long ldo = p.ld_out;
  const int b = (int)(s / sd), d = (int)(s % sd);
  int T = p.lengths ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  TOUT *out = (TOUT *)p.out + (size_t)b * Tmax * ldo;
  const int nw = ws.nw;

  constexpr int R = Q + 2;
  constexpr int kPF = prefetch_depth<Q>();
  double pend[Q + 1][Q + 1];
  double rp[Q + 1];
#pragma unroll
  for (int j = 0; j <= Q; ++j) {
    rp[j] = 0.0;
#pragma unroll
    for (int k = 0; k <= Q; ++k) pend[j][k] = 0.0;
  }
  auto load_rows = [&](double (&dst)[kPF][R], int f0) {
#pragma unroll
    for (int q = 0; q < kPF; ++q)
#pragma unroll
      for (int k = 0; k < R; ++k) dst[q][k] = (f0 + q < T) ? scratch[((size_t)(f0 + q) * R + k) * S + s] : 0.0;
  };

  int bad = 0;
  double cur[kPF][R], nxt[kPF][R];
  load_rows(cur, 0);
  for (int f0 = 0; f0 < T && !bad; f0 += kPF) {
    load_rows(nxt, f0 + kPF);
#pragma unroll
    for (int q = 0; q < kPF; ++q) {
      const int f = f0 + q;
      if (f >= T || bad) break;
      double v[Q + 1];
#pragma unroll
      for (int k = 0; k <= Q; ++k) v[k] = cur[q][k] + pend[0][k];
      if (v[0] <= 0.0) {  // NaN passes, as in linalg.pyx:78
        bad = f + 1;
        break;
      }
      const double iv0 = 1.0 / v[0];
      const double siv0 = sqrt(iv0);
      const double zf = (cur[q][Q + 1] + rp[0]) * siv0;
      double *sc = scratch + ((size_t)f * R) * S + s;
      sc[0] = siv0;
#pragma unroll
      for (int k = 1; k <= Q; ++k) {
        const double Lk = v[k] * siv0;
        sc[(size_t)k * S] = Lk;
        rp[k - 1] = rp[k] - Lk * zf;
      }
      sc[(size_t)(Q + 1) * S] = zf;
      // trailing update of the next Q columns (linalg.pyx:93-95), shifted one
      // column to the left so that pend[0] is always "the current frame"
#pragma unroll
      for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int l = 0; l <= Q; ++l) {
          double nv = pend[k + 1][l];
          if (l + k + 1 <= Q) nv -= v[l + k + 1] * v[k + 1] * iv0;
          pend[k][l] = nv;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kPF; ++q)
#pragma unroll
      for (int k = 0; k < R; ++k) cur[q][k] = nxt[q][k];
  }

  if (p.status) p.status[(size_t)b * p.ld_status + d] = bad;
  const int ncol = BWD ? nw : 1;
  if (bad) T = 0;  // failed system: zero-fill everything

  // reverse sweep: L^T x = z, rows prefetched kPF frames ahead (downwards)
  double xw[Q + 1];
#pragma unroll
  for (int k = 0; k <= Q; ++k) xw[k] = 0.0;
  auto load_rows_rev = [&](double (&dst)[kPF][R], int ftop) {  // frames ftop, ftop-1, ...
#pragma unroll
    for (int q = 0; q < kPF; ++q)
#pragma unroll
      for (int k = 0; k < R; ++k) dst[q][k] = (ftop - q >= 0 && ftop - q < T) ? scratch[((size_t)(ftop - q) * R + k) * S + s] : 0.0;
  };
  load_rows_rev(cur, T - 1);
  for (int f0 = T - 1; f0 >= 0; f0 -= kPF) {
    load_rows_rev(nxt, f0 - kPF);
#pragma unroll
    for (int q = 0; q < kPF; ++q) {
      const int f = f0 - q;
      if (f < 0) break;
      double x = cur[q][Q + 1];
#pragma unroll
      for (int k = 1; k <= Q; ++k) x -= cur[q][k] * xw[k];
      x *= cur[q][0];
#pragma unroll
      for (int k = Q; k >= 2; --k) xw[k] = xw[k - 1];
      if constexpr (Q >= 1) xw[1] = x;
      if (BWD)
        scratch[((size_t)f * R + (Q + 1)) * S + s] = x;
      else
        out[(size_t)f * ldo + d] = (TOUT)x;
    }
#pragma unroll
    for (int q = 0; q < kPF; ++q)
#pragma unroll
      for (int k = 0; k < R; ++k) cur[q][k] = nxt[q][k];
  }

  // zero the padding frames (and everything, for a failed system); the backward's valid frames are written
  // by the epilogue kernel
  for (int t = T; t < Tmax; ++t)
    for (int w = 0; w < ncol; ++w) out[(size_t)t * ldo + w * sd + d] = (TOUT)0;
}

// ---- stage 3 (backward only): grad[t, w*sd+d] = tau_w[t] * sum_k c_w[l_w+k] x[t+k], one thread per (frame, system) ----
template <int Q, typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void generic_epilogue_kernel(Problem p, WinSet ws, const double *__restrict__ scratch, long S) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= S * p.Tmax) return;
  const long s = e % S;
  const int t = (int)(e / S);
  const int sd = p.sd;
  const int b = (int)(s / sd), d = (int)(s % sd);
  int T = p.lengths ? p.lengths[b] : p.Tmax;
  T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
  if (t >= T) return;
  if (p.status && p.status[(size_t)b * p.ld_status + d] != 0) return;  // failed system: already zero-filled
  constexpr int R = Q + 2;
  const SysView<TIN, true> view = make_view<TIN, true>(p, ws, b, d, T);
  TOUT *out = (TOUT *)p.out + (size_t)b * p.Tmax * p.ld_out;
  for (int w = 0; w < ws.nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *c = ws.c + ws.off[w];
    double g = 0.0;
    for (int k = -l; k <= u; ++k) {
      const int tt = t + k;
      if (tt >= 0 && tt < T) g += c[l + k] * scratch[((size_t)tt * R + (Q + 1)) * S + s];
    }
    out[(size_t)t * p.ld_out + w * sd + d] = (TOUT)(view.tau(w, t) * g);
  }
}

template <int Q, typename TIN, typename TOUT, bool BWD>
int launch_q(hipStream_t st, const Problem &p, const WinSet &w, int device) {
  const long S = (long)p.B * p.sd;
  if (S == 0 || p.Tmax == 0) return 0;
  const size_t bytes = sizeof(double) * (size_t)(Q + 2) * (size_t)p.Tmax * (size_t)S;
  double *sc = (double *)scratch(device, st, 0, bytes);
  if (!sc) return MLPG_HIP_ENOMEM;
  Problem q = p;
  if (BWD && !q.status) {
    // the epilogue needs to know which systems failed even when the caller does not ask for the status
    q.status = (int32_t *)scratch(device, st, 2, sizeof(int32_t) * (size_t)S);
    if (!q.status) return MLPG_HIP_ENOMEM;
    q.ld_status = p.sd;
  }
  const long cells = S * p.Tmax;
  const unsigned gcells = (unsigned)((cells + 255) / 256);
  note_launch(kCountGeneric);
  hipLaunchKernelGGL((generic_assemble_kernel<Q, TIN, BWD>), dim3(gcells), dim3(256), 0, st, q, w, sc, S);
  const unsigned grid = (unsigned)((S + 63) / 64);
  hipLaunchKernelGGL((generic_kernel<Q, TIN, TOUT, BWD>), dim3(grid), dim3(64), 0, st, q, w, sc, S);
  if (BWD) hipLaunchKernelGGL((generic_epilogue_kernel<Q, TIN, TOUT>), dim3(gcells), dim3(256), 0, st, q, w, sc, S);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &w, int device) {
  if (w.q == 0) return launch_q<0, TIN, TOUT, BWD>(st, p, w, device);
  if (w.q == 1) return launch_q<1, TIN, TOUT, BWD>(st, p, w, device);
  if (w.q == 2) return launch_q<2, TIN, TOUT, BWD>(st, p, w, device);
  if (w.q <= 4) return launch_q<4, TIN, TOUT, BWD>(st, p, w, device);
  if (w.q <= 8) return launch_q<8, TIN, TOUT, BWD>(st, p, w, device);
  set_error("half-bandwidth %d > 8 is not supported", w.q);
  return MLPG_HIP_EINVAL;
}

// Delta features: out[b, t, w*D + d] = sum_k c_w[l_w + k] x[b, t + k, d]; one thread per (b, t, d),
// adjacent lanes = adjacent d (coalesced); the +-l/u neighbour rows come from L1/L2.
template <typename T>
__global__ void delta_kernel(const T *__restrict__ x, const int32_t *__restrict__ lengths, int B, int Tmax, int D,
                             WinSet ws, T *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Tmax * D;
  if (e >= total) return;
  const int d = (int)(e % D);
  const long bt = e / D;
  const int t = (int)(bt % Tmax), b = (int)(bt / Tmax);
  int len = lengths ? lengths[b] : Tmax;
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  const T *xb = x + (size_t)b * Tmax * D;
  T *ob = out + ((size_t)b * Tmax + t) * ((size_t)D * ws.nw);
  for (int w = 0; w < ws.nw; ++w) {
    double acc = 0.0;
    if (t < len) {
      const double *c = ws.c + ws.off[w];
      for (int k = -ws.l[w]; k <= ws.u[w]; ++k) {
        const int tt = t + k;
        if (tt >= 0 && tt < len) acc += c[ws.l[w] + k] * (double)xb[(size_t)tt * D + d];
      }
    }
    ob[(size_t)w * D + d] = (T)acc;
  }
}

// Pass-through of a stream without dynamic features: dst[b, t, c] = src[b, t, c] for t < len, else 0.
template <typename T>
__global__ void copy_cols_kernel(const T *__restrict__ src, long ld_src, const int32_t *__restrict__ lengths, int B,
                                 int Tmax, int ncols, T *__restrict__ dst, long ld_dst) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * Tmax * ncols) return;
  const int c = (int)(e % ncols);
  const long bt = e / ncols;
  const int t = (int)(bt % Tmax), b = (int)(bt / Tmax);
  int len = lengths ? lengths[b] : Tmax;
  len = len < 0 ? 0 : (len > Tmax ? Tmax : len);
  dst[(size_t)bt * ld_dst + c] = t < len ? src[(size_t)bt * ld_src + c] : (T)0;
}

// Plain streaming copy, 16 bytes per lane, ONE access per lane and a grid that covers the buffer: the yardstick for
// "what a copy kernel reaches on this box" (bench.py roofline.peak_measured), nothing else.  Measured on MI355X
// (tools/dbg/copy_sweep.hip, 512 MiB): this form 6.2 TB/s; persistent grid-stride loops of 1024 .. 16384 workgroups
// 4.4 - 5.4 TB/s (nontemporal or not); hipMemcpyAsync 5.1 TB/s -- the dispatcher walking the buffer front to back keeps
// the chip's accesses in a narrow moving window.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

}  // namespace

int launch_stream_copy(hipStream_t st, const void *src, void *dst, size_t nbytes) {
  const size_t n16 = nbytes / 16;
  if (n16 == 0) return 0;
  const size_t blocks = (n16 + 255) / 256;
  if (blocks > 0x7fffffffull) {
    set_error("stream_copy: buffer too large");
    return MLPG_HIP_EINVAL;
  }
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n16);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_copy_cols(hipStream_t st, int dtype, const void *src, long ld_src, const int32_t *lengths, int B, int Tmax,
                     int ncols, void *dst, long ld_dst) {
  const long total = (long)B * Tmax * ncols;
  if (total == 0) return 0;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == MLPG_HIP_F32)
    hipLaunchKernelGGL(copy_cols_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)src, ld_src, lengths, B,
                       Tmax, ncols, (float *)dst, ld_dst);
  else
    hipLaunchKernelGGL(copy_cols_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)src, ld_src, lengths,
                       B, Tmax, ncols, (double *)dst, ld_dst);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_delta(hipStream_t st, int dtype, const void *x, const int32_t *lengths, int B, int Tmax, int D,
                 const WinSet &w, void *out) {
  const long total = (long)B * Tmax * D;
  if (total == 0) return 0;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == MLPG_HIP_F32)
    hipLaunchKernelGGL(delta_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, lengths, B, Tmax, D, w,
                       (float *)out);
  else
    hipLaunchKernelGGL(delta_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x, lengths, B, Tmax, D, w,
                       (double *)out);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_generic(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w,
                   int device) {
  if (!backward) {
    return dtype == MLPG_HIP_F32 ? launch_t<float, float, false>(st, p, w, device)
                                 : launch_t<double, double, false>(st, p, w, device);
  }
  if (dtype == MLPG_HIP_F32)
    return out_dtype == MLPG_HIP_F32 ? launch_t<float, float, true>(st, p, w, device)
                                     : launch_t<float, double, true>(st, p, w, device);
  return out_dtype == MLPG_HIP_F32 ? launch_t<double, float, true>(st, p, w, device)
                                   : launch_t<double, double, true>(st, p, w, device);
}

}  // namespace mlpg
