// Modulation-spectrum kernels: the step after MLPG in the reference's
// pipelines.  Replace preprocessing/modspec.py (modspec :6-53, inv_modspec :62-100,
// modspec_smoothing :103-167: numpy rfft / irfft along the time axis of a (T, D) trajectory) and
// the Python loop over feature dimensions in autograd/_impl/modspec.py:30-60.
//
// One workgroup per (utterance, PAIR of adjacent feature columns): the two real columns are
// zero-padded to the DFT length n (a power of two <= 4096) and packed as real and imaginary part
// of ONE complex sequence z = x1 + i x2, transformed by a complex FFT that lives entirely in LDS
// (n points of 16 bytes + n/2 twiddles), separated (X1_k = (Z_k + conj Z_{n-k}) / 2,
// X2_k = (Z_k - conj Z_{n-k}) / 2i), modified, and -- for smoothing and for the backward --
// recombined and transformed back without leaving the chip.  HBM traffic is the trajectory in and
// the result out.
//
// FFT: in-place decimation in time on bit-reversed input, two radix-2 stages fused per LDS pass
// (a radix-4 butterfly in registers), twiddles exp(-2 pi i j / n) from sincospi in float64.
#include <math.h>

#include "common.h"

namespace mlpg {
namespace {

struct Cplx {
  double re, im;
};
__device__ __forceinline__ Cplx cadd(Cplx a, Cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cplx csub(Cplx a, Cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

constexpr int kFftThreads = 1024;

// In-place FFT of a[0..n) (already in bit-reversed order).  INV: conjugated twiddles (no scaling).
template <bool INV>
__device__ void fft_inplace(Cplx *a, const Cplx *tw, int n, int logn, int tid) {
  auto twid = [&](int idx) {
    Cplx w = tw[idx];
    if (INV) w.im = -w.im;
    return w;
  };
  int s = 0;
  if (logn & 1) {  // odd number of stages: one plain radix-2 stage first (twiddle 1)
    for (int b = tid; b < n / 2; b += kFftThreads) {
      const Cplx u = a[2 * b], v = a[2 * b + 1];
      a[2 * b] = cadd(u, v);
      a[2 * b + 1] = csub(u, v);
    }
    __syncthreads();
    s = 1;
  }
  for (; s < logn; s += 2) {  // stages s and s+1 in one pass
    const int h = 1 << s;
    for (int q = tid; q < n / 4; q += kFftThreads) {
      const int j = q & (h - 1);
      const int base = ((q >> s) << (s + 2)) | j;
      Cplx e0 = a[base], e1 = a[base + h], e2 = a[base + 2 * h], e3 = a[base + 3 * h];
      const Cplx w1 = twid(j << (logn - 1 - s));
      const Cplx t1 = cmul(e1, w1), t3 = cmul(e3, w1);
      const Cplx f0 = cadd(e0, t1), f1 = csub(e0, t1), f2 = cadd(e2, t3), f3 = csub(e2, t3);
      const Cplx wa = twid(j << (logn - 2 - s)), wb = twid((j + h) << (logn - 2 - s));
      const Cplx g2 = cmul(f2, wa), g3 = cmul(f3, wb);
      a[base] = cadd(f0, g2);
      a[base + 2 * h] = csub(f0, g2);
      a[base + h] = cadd(f1, g3);
      a[base + 3 * h] = csub(f1, g3);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int i, int logn) { return (int)(__brev((unsigned)i) >> (32 - logn)); }

enum { kModeSpec = 0, kModeInverse = 1, kModeSmooth = 2, kModeBackward = 3 };

struct ModArgs {
  const double *x;     // spec/smooth/backward: (B, T, D) trajectory
  const double *ms;    // inverse: (B, n/2+1, D) power spectrum; backward: gradient w.r.t. the power spectrum
  const double *ph;    // inverse: (B, n/2+1, D, 2) unit phasors
  double *out;         // spec: (B, n/2+1, D) power; inverse: (B, n, D); smooth/backward: (B, T, D)
  double *out_ph;      // spec: (B, n/2+1, D, 2) phasors or NULL
  int B, T, D, n, logn;
  int ortho;           // norm == "ortho"
  int limit_bin;       // smooth: first removed bin (> n/2: none)
  int log_domain;      // smooth: removed bins get unit magnitude (exp(0)) instead of zero
};

// spectra of the two packed real columns at bin k (0 <= k <= n/2) from Z_k and Z_{n-k}
__device__ __forceinline__ void unpack2(Cplx zk, Cplx zm, Cplx *x1, Cplx *x2) {
  *x1 = {0.5 * (zk.re + zm.re), 0.5 * (zk.im - zm.im)};
  *x2 = {0.5 * (zk.im + zm.im), 0.5 * (zm.re - zk.re)};
}
// Z_k and Z_{n-k} of z = h1 + i h2 for two HERMITIAN spectra given at bin k (their values at n-k are the conjugates)
__device__ __forceinline__ void pack2(Cplx h1, Cplx h2, Cplx *zk, Cplx *zm) {
  *zk = {h1.re - h2.im, h1.im + h2.re};
  *zm = {h1.re + h2.im, h2.re - h1.im};
}
__device__ __forceinline__ Cplx unit_phasor(Cplx s) {  // exp(i * angle(s)); numpy's angle(0) is 0
  const double mag = hypot(s.re, s.im);
  return mag > 0.0 ? Cplx{s.re / mag, s.im / mag} : Cplx{1.0, 0.0};
}

template <int MODE>
__global__ __launch_bounds__(kFftThreads) void modspec_kernel(ModArgs p) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx *a = (Cplx *)smem;
  Cplx *tw = a + p.n;
  const int tid = threadIdx.x;
  const int npair = (p.D + 1) / 2;
  const int d = 2 * (blockIdx.x % npair), b = blockIdx.x / npair;
  const bool two = d + 1 < p.D;  // the last pair of an odd D holds one column
  const int n = p.n, logn = p.logn, nb = n / 2 + 1, T = p.T, D = p.D;
  const double fwd_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0;
  const double inv_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0 / (double)n;

  for (int j = tid; j < n / 2; j += kFftThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)j / (double)n, &sn, &cs);
    tw[j] = {cs, sn};
  }

  if (MODE == kModeInverse) {
    // Hermitian spectra amp * phase of both columns (numpy's irfft ignores the imaginary part of bins 0 and
    // n/2), packed as H1 + i H2: one inverse transform returns column 1 in the real and column 2 in the
    // imaginary part
    const double *msb = p.ms + (size_t)b * nb * D + d;
    const double *phb = p.ph + ((size_t)b * nb * D + d) * 2;
    for (int k = tid; k < nb; k += kFftThreads) {
      const double a1 = sqrt(msb[(size_t)k * D]);
      Cplx h1 = {a1 * phb[(size_t)k * D * 2], a1 * phb[(size_t)k * D * 2 + 1]}, h2 = {0.0, 0.0};
      if (two) {
        const double a2 = sqrt(msb[(size_t)k * D + 1]);
        h2 = {a2 * phb[(size_t)k * D * 2 + 2], a2 * phb[(size_t)k * D * 2 + 3]};
      }
      if (k == 0 || k == n / 2) h1.im = h2.im = 0.0;
      Cplx zk, zm;
      pack2(h1, h2, &zk, &zm);
      a[bitrev(k, logn)] = zk;
      if (k != 0 && k != n / 2) a[bitrev(n - k, logn)] = zm;
    }
    __syncthreads();
    fft_inplace<true>(a, tw, n, logn, tid);
    double *ob = p.out + (size_t)b * n * D + d;
    for (int t = tid; t < n; t += kFftThreads) {
      ob[(size_t)t * D] = a[t].re * inv_scale;
      if (two) ob[(size_t)t * D + 1] = a[t].im * inv_scale;
    }
    return;
  }

  // forward transform of the two zero-padded columns
  const double *xb = p.x + (size_t)b * T * D + d;
  for (int t = tid; t < n; t += kFftThreads) {
    Cplx z = {0.0, 0.0};
    if (t < T) {
      z.re = xb[(size_t)t * D];
      if (two) z.im = xb[(size_t)t * D + 1];
    }
    a[bitrev(t, logn)] = z;
  }
  __syncthreads();
  fft_inplace<false>(a, tw, n, logn, tid);

  if (MODE == kModeSpec) {
    double *ob = p.out + (size_t)b * nb * D + d;
    for (int k = tid; k < nb; k += kFftThreads) {
      Cplx s1, s2;
      unpack2(a[k], a[(n - k) & (n - 1)], &s1, &s2);
      s1 = {s1.re * fwd_scale, s1.im * fwd_scale};
      s2 = {s2.re * fwd_scale, s2.im * fwd_scale};
      ob[(size_t)k * D] = s1.re * s1.re + s1.im * s1.im;
      if (two) ob[(size_t)k * D + 1] = s2.re * s2.re + s2.im * s2.im;
      if (p.out_ph) {
        double *pp = p.out_ph + ((size_t)b * nb * D + (size_t)k * D + d) * 2;
        const Cplx u1 = unit_phasor(s1);
        pp[0] = u1.re;
        pp[1] = u1.im;
        if (two) {
          const Cplx u2 = unit_phasor(s2);
          pp[2] = u2.re;
          pp[3] = u2.im;
        }
      }
    }
    return;
  }

  // both remaining modes rebuild Z' = H1 + i H2 from per-column Hermitian spectra, one thread per bin pair (k, n-k)
  const double *gb = MODE == kModeBackward ? p.ms + (size_t)b * nb * D + d : nullptr;
  for (int k = tid; k < nb; k += kFftThreads) {
    const int km = (n - k) & (n - 1);
    Cplx s1, s2;
    unpack2(a[k], a[km], &s1, &s2);
    s1 = {s1.re * fwd_scale, s1.im * fwd_scale};
    s2 = {s2.re * fwd_scale, s2.im * fwd_scale};
    Cplx h1, h2;
    if (MODE == kModeSmooth) {
      // bins >= limit_bin: power := 0, or log-power := 0 (unit magnitude, phase kept) in the log domain
      h1 = s1;
      h2 = s2;
      if (k >= p.limit_bin) {
        h1 = p.log_domain ? unit_phasor(s1) : Cplx{0.0, 0.0};
        h2 = p.log_domain ? unit_phasor(s2) : Cplx{0.0, 0.0};
      }
      if (k == 0 || k == n / 2) h1.im = h2.im = 0.0;
    } else {
      // grad[t] = C Re sum_{k <= n/2} g_k S_k e^{+2 pi i k t / n}: as a Hermitian spectrum, g_k S_k / 2 at
      // 0 < k < n/2 (and its conjugate at n-k), Re(g_k S_k) at k = 0 and n/2
      const double g1 = gb[(size_t)k * D], g2 = two ? gb[(size_t)k * D + 1] : 0.0;
      const bool edge = k == 0 || k == n / 2;
      const double f = edge ? 1.0 : 0.5;
      h1 = {f * g1 * s1.re, edge ? 0.0 : f * g1 * s1.im};
      h2 = {f * g2 * s2.re, edge ? 0.0 : f * g2 * s2.im};
    }
    if (!two) h2 = {0.0, 0.0};
    Cplx zk, zm;
    pack2(h1, h2, &zk, &zm);
    a[k] = zk;
    if (km != k) a[km] = zm;
  }
  __syncthreads();
  // the inverse transform wants bit-reversed input: permute in place (swap pairs)
  for (int k = tid; k < n; k += kFftThreads) {
    const int r = bitrev(k, logn);
    if (r > k) {
      const Cplx t = a[k];
      a[k] = a[r];
      a[r] = t;
    }
  }
  __syncthreads();
  fft_inplace<true>(a, tw, n, logn, tid);
  double *ob = p.out + (size_t)b * T * D + d;
  // smoothing: irfft scaling; backward: C = 2 (2 / sqrt(n) with "ortho"), autograd/_impl/modspec.py:47-49
  const double osc = MODE == kModeSmooth ? inv_scale : (p.ortho ? 2.0 / sqrt((double)n) : 2.0);
  for (int t = tid; t < T; t += kFftThreads) {
    ob[(size_t)t * D] = a[t].re * osc;
    if (two) ob[(size_t)t * D + 1] = a[t].im * osc;
  }
}

template <int MODE>
int launch_mode(hipStream_t st, const ModArgs &p) {
  const size_t lds = sizeof(Cplx) * ((size_t)p.n + (size_t)p.n / 2);
  auto kern = modspec_kernel<MODE>;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * ((p.D + 1) / 2))), dim3(kFftThreads), lds, st, p);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace

int launch_modspec(hipStream_t st, int mode, const double *x, const double *ms, const double *ph, double *out,
                   double *out_ph, int B, int T, int D, int n, int ortho, int limit_bin, int log_domain) {
  int logn = 0;
  while ((1 << logn) < n) ++logn;
  if (n < 2 || n > 4096 || (1 << logn) != n) {
    set_error("modspec: the DFT length must be a power of two in [2, 4096] (got %d)", n);
    return MLPG_HIP_EINVAL;
  }
  ModArgs p;
  p.x = x; p.ms = ms; p.ph = ph; p.out = out; p.out_ph = out_ph;
  p.B = B; p.T = T; p.D = D; p.n = n; p.logn = logn;
  p.ortho = ortho; p.limit_bin = limit_bin; p.log_domain = log_domain;
  switch (mode) {
    case kModeSpec: return launch_mode<kModeSpec>(st, p);
    case kModeInverse: return launch_mode<kModeInverse>(st, p);
    case kModeSmooth: return launch_mode<kModeSmooth>(st, p);
    case kModeBackward: return launch_mode<kModeBackward>(st, p);
  }
  set_error("modspec: bad mode %d", mode);
  return MLPG_HIP_EINVAL;
}

}  // namespace mlpg
