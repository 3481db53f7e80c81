// Modulation-spectrum kernels: the step after MLPG in the reference's
// pipelines.  Replace preprocessing/modspec.py (modspec :6-53, inv_modspec :62-100,
// modspec_smoothing :103-167: numpy rfft / irfft along the time axis of a (T, D) trajectory) and
// the Python loop over feature dimensions in autograd/_impl/modspec.py:30-60.
//
// One workgroup per (utterance, PAIR of adjacent feature columns): the two real columns are
// zero-padded to the DFT length n (a power of two <= 4096) and packed as real and imaginary part
// of ONE complex sequence z = x1 + i x2, transformed by a complex FFT that lives entirely in LDS
// (n points of 16 bytes + per-pass twiddle tables), separated (X1_k = (Z_k + conj Z_{n-k}) / 2,
// X2_k = (Z_k - conj Z_{n-k}) / 2i), modified, and -- for smoothing and for the backward --
// recombined and transformed back without leaving the chip.  HBM traffic is the trajectory in and
// the result out.
//
// FFT: in-place decimation in time on bit-reversed input.  First pass: a radix-16 (radix-8 for odd
// log2 n) transform of 16 consecutive elements in registers; then radix-4 passes (two radix-2
// stages fused).  The data are padded by one slot per 16 elements and every pass has its own compact
// twiddle table (sincospi, float64), so that no LDS access of the transform has a bank conflict
// by construction.
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

// LDS layout: element i lives at a[pidx(i)], one padding slot per 16 elements, so that both the
// first pass (every thread owns 8 or 16 CONSECUTIVE elements) and the later passes (consecutive
// threads touch consecutive elements) are free of bank conflicts.
__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }
constexpr int padded_len(int n) { return n + (n >> 4) + 1; }

// Twiddles: one compact table per radix-4 pass (stages s, s+1; h = 2^s): tab[j] = W_{4h}^j, j < 2h,
// read by consecutive threads at consecutive addresses.  Pass tables are stored back to back;
// tw_offset(s0, s) = 2 * (h(s0) + h(s0 + 2) + ... below s) entries.
__device__ __forceinline__ int tw_offset(int s0, int s) {
  int off = 0;
  for (int t = s0; t < s; t += 2) off += 2 << t;
  return off;
}

// In-register DIT FFT of R = 2^LOGR consecutive elements (input in bit-reversed order), W_R = exp(-+ 2 pi i / R)
template <int LOGR, bool INV>
__device__ __forceinline__ void fft_regs(Cplx (&v)[1 << LOGR]) {
  constexpr int R = 1 << LOGR;
  // cos / sin of 2 pi k / 16, k = 0..7
  constexpr double c16[8] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173,
                             0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848};
  constexpr double s16[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848,
                             1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173};
#pragma unroll
  for (int t = 0; t < LOGR; ++t) {
    const int h = 1 << t;
#pragma unroll
    for (int b = 0; b < R / 2; ++b) {
      const int j = b & (h - 1);
      const int i0 = ((b >> t) << (t + 1)) | j;
      const int k16 = j * (8 >> t);  // W_{2h}^j = W_16^{j * 16 / (2h)}
      const Cplx w = {c16[k16], INV ? s16[k16] : -s16[k16]};
      const Cplx u = v[i0], x = (k16 == 0) ? v[i0 + h] : cmul(v[i0 + h], w);
      v[i0] = cadd(u, x);
      v[i0 + h] = csub(u, x);
    }
  }
}

// In-place FFT of the n elements at a[pidx(.)] (already in bit-reversed order).  INV: conjugated
// twiddles (no scaling).  First pass: radix 16 (radix 8 when log2 n is odd) in registers; then
// radix-4 passes.  tw: the per-pass tables described above (built by build_twiddles).
template <bool INV>
__device__ void fft_inplace(Cplx *a, const Cplx *tw, int n, int logn, int tid) {
  int s0;
  if (logn < 3) {  // n = 2 or 4: plain radix-2 stages by one thread each
    for (int t = 0; t < logn; ++t) {
      const int h = 1 << t;
      for (int b = tid; b < n / 2; b += kFftThreads) {
        const int j = b & (h - 1), i0 = ((b >> t) << (t + 1)) | j;
        Cplx w = {1.0, 0.0};
        if (t == 1 && j == 1) w = {0.0, INV ? 1.0 : -1.0};
        const Cplx u = a[pidx(i0)], x = cmul(a[pidx(i0 + h)], w);
        a[pidx(i0)] = cadd(u, x);
        a[pidx(i0 + h)] = csub(u, x);
      }
      __syncthreads();
    }
    return;
  }
  if (logn & 1) {
    for (int q = tid; q < n / 8; q += kFftThreads) {
      Cplx v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = a[pidx(8 * q + k)];
      fft_regs<3, INV>(v);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[pidx(8 * q + k)] = v[k];
    }
    s0 = 3;
  } else {
    for (int q = tid; q < n / 16; q += kFftThreads) {
      Cplx v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = a[pidx(16 * q + k)];
      fft_regs<4, INV>(v);
#pragma unroll
      for (int k = 0; k < 16; ++k) a[pidx(16 * q + k)] = v[k];
    }
    s0 = 4;
  }
  __syncthreads();
  for (int s = s0; s < logn; s += 2) {  // stages s and s+1 in one pass
    const int h = 1 << s;
    const Cplx *tab = tw + tw_offset(s0, s);  // W_{4h}^j, j < 2h
    for (int q = tid; q < n / 4; q += kFftThreads) {
      const int j = q & (h - 1);
      const int base = ((q >> s) << (s + 2)) | j;
      const int p0 = pidx(base), p1 = pidx(base + h), p2 = pidx(base + 2 * h), p3 = pidx(base + 3 * h);
      Cplx e0 = a[p0], e1 = a[p1], e2 = a[p2], e3 = a[p3];
      Cplx w1 = tab[2 * j], wa = tab[j], wb = tab[j + h];  // W_{2h}^j = W_{4h}^{2j}
      if (INV) {
        w1.im = -w1.im;
        wa.im = -wa.im;
        wb.im = -wb.im;
      }
      const Cplx t1 = cmul(e1, w1), t3 = cmul(e3, w1);
      const Cplx f0 = cadd(e0, t1), f1 = csub(e0, t1), f2 = cadd(e2, t3), f3 = csub(e2, t3);
      const Cplx g2 = cmul(f2, wa), g3 = cmul(f3, wb);
      a[p0] = cadd(f0, g2);
      a[p2] = csub(f0, g2);
      a[p1] = cadd(f1, g3);
      a[p3] = csub(f1, g3);
    }
    __syncthreads();
  }
}

// forward twiddles of every radix-4 pass of an n-point transform (see tw_offset); < n entries in total
__device__ void build_twiddles(Cplx *tw, int logn, int tid) {
  if (logn < 3) return;
  const int s0 = (logn & 1) ? 3 : 4;
  for (int s = s0; s < logn; s += 2) {
    const int h = 1 << s;
    Cplx *tab = tw + tw_offset(s0, s);
    for (int j = tid; j < 2 * h; j += kFftThreads) {
      double sn, cs;
      sincospi(-(double)j / (double)(2 * h), &sn, &cs);  // -2 pi j / (4h)
      tab[j] = {cs, sn};
    }
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
  Cplx *tw = a + padded_len(p.n);
  const int tid = threadIdx.x;
  const int npair = (p.D + 1) / 2;
  const int d = 2 * (blockIdx.x % npair), b = blockIdx.x / npair;
  const bool two = d + 1 < p.D;  // the last pair of an odd D holds one column
  const int n = p.n, logn = p.logn, nb = n / 2 + 1, T = p.T, D = p.D;
  const double fwd_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0;
  const double inv_scale = p.ortho ? 1.0 / sqrt((double)n) : 1.0 / (double)n;

  build_twiddles(tw, logn, tid);

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
      a[pidx(bitrev(k, logn))] = zk;
      if (k != 0 && k != n / 2) a[pidx(bitrev(n - k, logn))] = zm;
    }
    __syncthreads();
    fft_inplace<true>(a, tw, n, logn, tid);
    double *ob = p.out + (size_t)b * n * D + d;
    for (int t = tid; t < n; t += kFftThreads) {
      ob[(size_t)t * D] = a[pidx(t)].re * inv_scale;
      if (two) ob[(size_t)t * D + 1] = a[pidx(t)].im * inv_scale;
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
    a[pidx(bitrev(t, logn))] = z;
  }
  __syncthreads();
  fft_inplace<false>(a, tw, n, logn, tid);

  if (MODE == kModeSpec) {
    double *ob = p.out + (size_t)b * nb * D + d;
    for (int k = tid; k < nb; k += kFftThreads) {
      Cplx s1, s2;
      unpack2(a[pidx(k)], a[pidx((n - k) & (n - 1))], &s1, &s2);
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
    unpack2(a[pidx(k)], a[pidx(km)], &s1, &s2);
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
    a[pidx(k)] = zk;
    if (km != k) a[pidx(km)] = zm;
  }
  __syncthreads();
  // the inverse transform wants bit-reversed input: permute in place (swap pairs)
  for (int k = tid; k < n; k += kFftThreads) {
    const int r = bitrev(k, logn);
    if (r > k) {
      const Cplx t = a[pidx(k)];
      a[pidx(k)] = a[pidx(r)];
      a[pidx(r)] = t;
    }
  }
  __syncthreads();
  fft_inplace<true>(a, tw, n, logn, tid);
  double *ob = p.out + (size_t)b * T * D + d;
  // smoothing: irfft scaling; backward: C = 2 (2 / sqrt(n) with "ortho"), autograd/_impl/modspec.py:47-49
  const double osc = MODE == kModeSmooth ? inv_scale : (p.ortho ? 2.0 / sqrt((double)n) : 2.0);
  for (int t = tid; t < T; t += kFftThreads) {
    ob[(size_t)t * D] = a[pidx(t)].re * osc;
    if (two) ob[(size_t)t * D + 1] = a[pidx(t)].im * osc;
  }
}

template <int MODE>
int launch_mode(hipStream_t st, const ModArgs &p) {
  const size_t lds = sizeof(Cplx) * ((size_t)padded_len(p.n) + (size_t)p.n);  // data + per-pass twiddle tables (< n entries)
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
