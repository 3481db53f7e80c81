// Modulation spectrum for DFT lengths the in-LDS FFT of modspec.hip does not take: any n that is not a power of
// two, or n > 4096 (numpy's rfft / irfft, which the reference calls at preprocessing/modspec.py:45,91,151-164 and
// autograd/_impl/modspec.py:30-60, accept every n).  Same four modes, same results; the transform is the DFT sum
// itself, O(n) per output value:
//
//   forward:  S[b, k, d] = sum_{t < min(T, n)} x[b, t, d] e^{-2 pi i k t / n},    0 <= k <= n/2
//   inverse:  y[b, t, d] = scale * sum_{k <= n/2} c_k Re(H[b, k, d] e^{+2 pi i k t / n})
//             (c_k = 2 except 1 at k = 0 and, for even n, k = n/2: the Hermitian completion irfft applies;
//              c_k = 1 for the gradient, autograd/_impl/modspec.py:47-60)
//
// Both kernels tile (64 outputs) x (16 feature columns) per workgroup and walk the summation index through LDS
// tiles; the phase k t mod n is kept as an exact integer and looked up in a table of the n-th roots of unity
// (float64 sincospi), so the twiddles carry no accumulated error however long the transform.  The half spectrum
// between the two kernels (smoothing, backward) lives in stream scratch.  This is the completeness path: ~25 ms for
// 256 x 60 columns at n = 5000, against ~1 s for numpy on one core; the power-of-two lengths the reference's
// defaults use (2048, 4096) stay on the fused FFT kernel.
#include <math.h>

#include "common.h"

namespace mlpg {
namespace {

enum { kModeSpec = 0, kModeInverse = 1, kModeSmooth = 2, kModeBackward = 3 };
constexpr int kTile = 64, kCols = 16;

struct DftArgs {
  const double *x;    // (B, T, D)
  const double *ms;   // inverse: power (B, nb, D); backward: gradient w.r.t. the power
  const double *ph;   // inverse: unit phasors (B, nb, D, 2)
  double *out;
  double *out_ph;
  double2 *H;         // (B, nb, D) half spectrum between the two kernels
  const double2 *tw;  // tw[j] = (cos, sin)(2 pi j / n)
  int B, T, D, n, nb;
  int ortho, limit_bin, log_domain;
};

__global__ void dft_twiddles(double2 *tw, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s, c;
  sincospi(2.0 * (double)j / (double)n, &s, &c);
  tw[j] = make_double2(c, s);
}

__device__ __forceinline__ double2 unit_phasor(double2 s) {  // exp(i angle(s)); numpy's angle(0) is 0
  const double mag = hypot(s.x, s.y);
  return mag > 0.0 ? make_double2(s.x / mag, s.y / mag) : make_double2(1.0, 0.0);
}

// grid (ceil(nb / 64), ceil(D / 16), B), 256 threads: thread = (bin k0 + (tid & 63), columns d0 + 4 (tid >> 6) ..+3)
template <int MODE>
__global__ __launch_bounds__(256) void dft_forward(DftArgs p) {
  __shared__ double xs[kTile][kCols];
  const int tid = threadIdx.x, kk = tid & 63, cq = tid >> 6;
  const int k = blockIdx.x * kTile + kk, d0 = blockIdx.y * kCols, b = blockIdx.z;
  const int n = p.n, nb = p.nb, D = p.D, T = p.T;
  const int Tn = T < n ? T : n;
  const int kc = k < nb ? k : nb - 1;  // idle threads shadow the last bin
  double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
  int j = 0;  // k t mod n
  const double *xb = p.x + (size_t)b * T * D;
  for (int t0 = 0; t0 < Tn; t0 += kTile) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, tt = idx >> 4, c = idx & 15;
      xs[tt][c] = (t0 + tt < Tn && d0 + c < D) ? xb[(size_t)(t0 + tt) * D + d0 + c] : 0.0;
    }
    __syncthreads();
    const int lim = Tn - t0 < kTile ? Tn - t0 : kTile;
    for (int tt = 0; tt < lim; ++tt) {
      const double2 w = p.tw[j];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double xv = xs[tt][cq * 4 + c];
        re[c] += xv * w.x;
        im[c] -= xv * w.y;
      }
      j += kc;
      if (j >= n) j -= n;
    }
    __syncthreads();
  }
  if (k >= nb) return;
  const double fwd_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = d0 + cq * 4 + c;
    if (d >= D) break;
    const size_t o = ((size_t)b * nb + k) * D + d;
    const double2 s = make_double2(re[c] * fwd_scale, im[c] * fwd_scale);
    if (MODE == kModeSpec) {
      p.out[o] = s.x * s.x + s.y * s.y;
      if (p.out_ph) {
        const double2 u = unit_phasor(s);
        p.out_ph[2 * o] = u.x;
        p.out_ph[2 * o + 1] = u.y;
      }
    } else if (MODE == kModeSmooth) {
      // bins >= limit_bin: power := 0, or log-power := 0 (unit magnitude, phase kept) in the log domain
      double2 h = s;
      if (k >= p.limit_bin) h = p.log_domain ? unit_phasor(s) : make_double2(0.0, 0.0);
      p.H[o] = h;
    } else {  // backward: g_k S_k
      const double g = p.ms[o];
      p.H[o] = make_double2(g * s.x, g * s.y);
    }
  }
}

// grid (ceil(Tout / 64), ceil(D / 16), B): thread = (sample t0 + (tid & 63), columns d0 + 4 (tid >> 6) ..+3)
template <int MODE>
__global__ __launch_bounds__(256) void dft_inverse(DftArgs p, int Tout, double scale) {
  __shared__ double2 hs[kTile][kCols];
  const int tid = threadIdx.x, tl = tid & 63, cq = tid >> 6;
  const int t = blockIdx.x * kTile + tl, d0 = blockIdx.y * kCols, b = blockIdx.z;
  const int n = p.n, nb = p.nb, D = p.D;
  const int tc = (t < Tout ? t : Tout - 1) % n;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int j = 0;  // k t mod n
  for (int k0 = 0; k0 < nb; k0 += kTile) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, kk = idx >> 4, c = idx & 15;
      const int k = k0 + kk, d = d0 + c;
      double2 h = make_double2(0.0, 0.0);
      if (k < nb && d < D) {
        const size_t o = ((size_t)b * nb + k) * D + d;
        if (MODE == kModeInverse) {
          const double a = sqrt(p.ms[o]);
          h = make_double2(a * p.ph[2 * o], a * p.ph[2 * o + 1]);
        } else {
          h = p.H[o];
        }
        // Hermitian completion (irfft): every bin counts twice except 0 and n/2; the gradient sums them once
        const bool once = MODE == kModeBackward || k == 0 || (!(n & 1) && k == n / 2);
        if (!once) h = make_double2(2.0 * h.x, 2.0 * h.y);
      }
      hs[kk][c] = h;
    }
    __syncthreads();
    const int lim = nb - k0 < kTile ? nb - k0 : kTile;
    for (int kk = 0; kk < lim; ++kk) {
      const double2 w = p.tw[j];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 h = hs[kk][cq * 4 + c];
        acc[c] += h.x * w.x - h.y * w.y;
      }
      j += tc;
      if (j >= n) j -= n;
    }
    __syncthreads();
  }
  if (t >= Tout) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = d0 + cq * 4 + c;
    if (d >= D) break;
    p.out[((size_t)b * Tout + t) * D + d] = acc[c] * scale;
  }
}

}  // namespace

int launch_modspec_dft(hipStream_t st, int device, int mode, const double *x, const double *ms, const double *ph,
                       double *out, double *out_ph, int B, int T, int D, int n, int ortho, int limit_bin,
                       int log_domain) {
  if (B > 65535) {
    set_error("modspec: more than 65535 sequences per call with a DFT length that is not a power of two <= 4096");
    return MLPG_HIP_EINVAL;
  }
  DftArgs p;
  p.x = x; p.ms = ms; p.ph = ph; p.out = out; p.out_ph = out_ph;
  p.B = B; p.T = T; p.D = D; p.n = n; p.nb = n / 2 + 1;
  p.ortho = ortho; p.limit_bin = limit_bin; p.log_domain = log_domain;
  const bool two_pass = mode == kModeSmooth || mode == kModeBackward;
  const size_t tw_bytes = ((size_t)n * sizeof(double2) + 255) / 256 * 256;
  const size_t h_bytes = two_pass ? (size_t)B * p.nb * D * sizeof(double2) : 0;
  char *sc = (char *)scratch(device, st, 0, tw_bytes + h_bytes);
  if (!sc) return MLPG_HIP_ENOMEM;
  p.tw = (const double2 *)sc;
  p.H = (double2 *)(sc + tw_bytes);
  hipLaunchKernelGGL(dft_twiddles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (double2 *)sc, n);
  MLPG_HIP_CHECK(hipGetLastError());
  const dim3 blk(256);
  const dim3 gf((unsigned)((p.nb + kTile - 1) / kTile), (unsigned)((D + kCols - 1) / kCols), (unsigned)B);
  const double inv_scale = ortho ? 1.0 / sqrt((double)n) : 1.0 / (double)n;
  switch (mode) {
    case kModeSpec:
      hipLaunchKernelGGL(dft_forward<kModeSpec>, gf, blk, 0, st, p);
      break;
    case kModeInverse: {
      const dim3 gi((unsigned)((n + kTile - 1) / kTile), gf.y, gf.z);
      hipLaunchKernelGGL(dft_inverse<kModeInverse>, gi, blk, 0, st, p, n, inv_scale);
      break;
    }
    case kModeSmooth: {
      if (T == 0) return 0;
      hipLaunchKernelGGL(dft_forward<kModeSmooth>, gf, blk, 0, st, p);
      MLPG_HIP_CHECK(hipGetLastError());
      const dim3 gi((unsigned)((T + kTile - 1) / kTile), gf.y, gf.z);
      hipLaunchKernelGGL(dft_inverse<kModeSmooth>, gi, blk, 0, st, p, T, inv_scale);
      break;
    }
    case kModeBackward: {
      if (T == 0) return 0;
      hipLaunchKernelGGL(dft_forward<kModeBackward>, gf, blk, 0, st, p);
      MLPG_HIP_CHECK(hipGetLastError());
      const dim3 gi((unsigned)((T + kTile - 1) / kTile), gf.y, gf.z);
      // C = 2 (2 / sqrt(n) with "ortho"), autograd/_impl/modspec.py:47-49
      hipLaunchKernelGGL(dft_inverse<kModeBackward>, gi, blk, 0, st, p, T, ortho ? 2.0 / sqrt((double)n) : 2.0);
      break;
    }
    default:
      set_error("modspec: bad mode %d", mode);
      return MLPG_HIP_EINVAL;
  }
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mlpg
