// Modulation-spectrum kernels: the step after MLPG in the reference's
// pipelines.  Replace preprocessing/modspec.py (modspec :6-53, inv_modspec :62-100,
// modspec_smoothing :103-167: numpy rfft / irfft along the time axis of a (T, D) trajectory) and
// the Python loop over feature dimensions in autograd/_impl/modspec.py:30-60.
//
// One workgroup per (utterance, feature column): the column is zero-padded to the DFT length n
// (a power of two <= 4096), transformed by a complex FFT that lives entirely in LDS (n points of
// 16 bytes + n/2 twiddles), modified, and -- for smoothing and for the backward -- transformed
// back without leaving the chip.  HBM traffic is the trajectory in and the result out.
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

constexpr int kFftThreads = 256;

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

template <int MODE>
__global__ __launch_bounds__(kFftThreads) void modspec_kernel(ModArgs p) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx *a = (Cplx *)smem;
  Cplx *tw = a + p.n;
  const int tid = threadIdx.x;
  const int d = blockIdx.x % p.D, b = blockIdx.x / p.D;
  const int n = p.n, logn = p.logn, nb = n / 2 + 1, T = p.T, D = p.D;
  const double fwd_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0;
  const double inv_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0 / (double)n;

  for (int j = tid; j < n / 2; j += kFftThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)j / (double)n, &sn, &cs);
    tw[j] = {cs, sn};
  }

  if (MODE == kModeInverse) {
    // Hermitian extension of amp * phase (numpy's irfft ignores the imaginary part of bins 0 and n/2)
    const double *msb = p.ms + (size_t)b * nb * D + d;
    const double *phb = p.ph + ((size_t)b * nb * D + d) * 2;
    for (int k = tid; k < n; k += kFftThreads) {
      const int kk = k <= n / 2 ? k : n - k;
      const double amp = sqrt(msb[(size_t)kk * D]);
      Cplx z = {amp * phb[(size_t)kk * D * 2], amp * phb[(size_t)kk * D * 2 + 1]};
      if (k > n / 2) z.im = -z.im;
      if (kk == 0 || kk == n / 2) z.im = 0.0;
      a[bitrev(k, logn)] = z;
    }
    __syncthreads();
    fft_inplace<true>(a, tw, n, logn, tid);
    double *ob = p.out + (size_t)b * n * D + d;
    for (int t = tid; t < n; t += kFftThreads) ob[(size_t)t * D] = a[t].re * inv_scale;
    return;
  }

  // forward transform of the zero-padded column
  const double *xb = p.x + (size_t)b * T * D + d;
  for (int t = tid; t < n; t += kFftThreads) a[bitrev(t, logn)] = {t < T ? xb[(size_t)t * D] : 0.0, 0.0};
  __syncthreads();
  fft_inplace<false>(a, tw, n, logn, tid);

  if (MODE == kModeSpec) {
    double *ob = p.out + (size_t)b * nb * D + d;
    for (int k = tid; k < nb; k += kFftThreads) {
      const double re = a[k].re * fwd_scale, im = a[k].im * fwd_scale;
      ob[(size_t)k * D] = re * re + im * im;
      if (p.out_ph) {
        // exp(i * angle(s)): s / |s|, and 1 for s == 0 (numpy's angle(0) is 0)
        const double mag = hypot(re, im);
        double *pp = p.out_ph + ((size_t)b * nb * D + (size_t)k * D + d) * 2;
        pp[0] = mag > 0.0 ? re / mag : 1.0;
        pp[1] = mag > 0.0 ? im / mag : 0.0;
      }
    }
    return;
  }

  if (MODE == kModeSmooth) {
    // bins >= limit_bin: power := 0, or log-power := 0 (unit magnitude, phase kept) in the log domain
    for (int k = tid; k < n; k += kFftThreads) {
      const int kk = k <= n / 2 ? k : n - k;
      Cplx z = a[k];
      z.re *= fwd_scale;
      z.im *= fwd_scale;
      if (kk >= p.limit_bin) {
        if (p.log_domain) {
          const double mag = hypot(z.re, z.im);
          z = mag > 0.0 ? Cplx{z.re / mag, z.im / mag} : Cplx{1.0, 0.0};
          if (mag == 0.0 && k > n / 2) z.im = -z.im;
        } else {
          z = {0.0, 0.0};
        }
      }
      if (kk == 0 || kk == n / 2) z.im = 0.0;
      a[k] = z;
    }
  } else {  // kModeBackward: one-sided spectrum g_k * S_k
    const double *gb = p.ms + (size_t)b * nb * D + d;
    for (int k = tid; k < n; k += kFftThreads) {
      Cplx z = {0.0, 0.0};
      if (k <= n / 2) {
        const double g = gb[(size_t)k * D] * fwd_scale;
        z = {a[k].re * g, a[k].im * g};
      }
      a[k] = z;
    }
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
  if (MODE == kModeSmooth) {
    for (int t = tid; t < T; t += kFftThreads) ob[(size_t)t * D] = a[t].re * inv_scale;
  } else {
    // grad[t] = C * Re sum_{k <= n/2} g_k S_k e^{+2 pi i k t / n},  C = 2 (2 / sqrt(n) with "ortho")
    const double C = p.ortho ? 2.0 / sqrt((double)n) : 2.0;
    for (int t = tid; t < T; t += kFftThreads) ob[(size_t)t * D] = C * a[t].re;
  }
}

template <int MODE>
int launch_mode(hipStream_t st, const ModArgs &p) {
  const size_t lds = sizeof(Cplx) * ((size_t)p.n + (size_t)p.n / 2);
  auto kern = modspec_kernel<MODE>;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.D)), dim3(kFftThreads), lds, st, p);
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
