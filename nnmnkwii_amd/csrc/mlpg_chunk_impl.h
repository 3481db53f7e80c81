// MLPG forward for window extents up to 2 (MLPG_HIP_ALGO_CHUNK): P = sum_w W_w^T diag(tau_w) W_w has half-bandwidth
// Q = 2 * extent <= 4 (the reference's own 5-tap test windows, tests/test_paramgen.py:21-26; paramgen/_mlpg.py:92-199 solves
// them with the same banded Cholesky as any other window set).  Executable specification: tools/chunk_model.py.
//
// Lane = static dim (coalesced rows, as the strip kernel), wavefront = one chunk of C = I + Q frames: I interior frames
// followed by a separator of Q frames.  Three passes, none of which waits for another workgroup:
//   pass 1  chunk_kernel<.., P3 = false>: the chunk's rows of P are assembled on the fly (every frame loaded once, its
//           precisions spread over the 2 * EXT + 1 rows it touches) and eliminated in natural order with the INTERIOR rows as
//           the only pivots; the Q columns that couple the first interior rows to the previous chunk's separator ride along
//           as right-hand sides.  Only the chunk's record survives (Q = 4: 44 doubles per lane).
//   pass 2  reduce_elim_kernel + reduce_subst_kernel: the separators' block-tridiagonal system (Q x Q blocks), two wavefronts per
//           (utterance, dim group): elimination from both ends towards the middle, substitution from the middle outwards: the
//           separators' solutions.
//   pass 3  chunk_kernel<.., P3 = true>: the same elimination AGAIN with the neighbouring separators' solutions known; the
//           factor rows of the chunk stay on chip this time (the first NLDS rows in LDS, the others in registers),
//           back-substitution, the trajectory is stored.
// A non-positive pivot anywhere marks the lane; verdict_kernel gives marked systems the reference's verdict (natural-order
// first failing pivot, linalg.pyx:79-82) and a zero column.
// The price of not waiting is reading the inputs twice.
#pragma once
#include <type_traits>

#include "assemble.h"

namespace mlpg {
namespace chunk {

constexpr int kW = 4;      // wavefronts (chunks) per workgroup
constexpr int kMaxNw = 3;  // windows this kernel takes (the reference ships static + delta + delta-delta sets)

template <int Q>
struct Geo {
  static constexpr int EXT = Q / 2;
  static constexpr int I = Q == 4 ? 16 : 14;
  static constexpr int C = I + Q;
  static constexpr int NLDS = Q == 4 ? 8 : 0;  // rows of pass 3's factor that live in LDS
  static constexpr int NS = Q * (Q + 1) / 2;   // a symmetric Q x Q block, packed
  static constexpr int kRec = 2 * NS + Q * Q + 2 * Q;
  static constexpr int oSLL = 0, oSRR = NS, oSRL = 2 * NS, oGL = 2 * NS + Q * Q, oGR = 2 * NS + Q * Q + Q;
  static constexpr int kFac = NS + Q;          // unit-lower multipliers (NS - Q) + inverse pivots (Q) + right-hand side (Q)
  static constexpr int NP = (2 * EXT + 1) * (2 * EXT + 2) / 2;  // coefficient products per window
};
__host__ __device__ constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }  // a >= b

struct Args {
  double *rec;  // [g][chunk][kRec][64]
  double *fac;  // [g][chunk][kFac][64]
  double *xs;   // [g][chunk][Q][64]
  double *mid;  // [g][kFac][64]: the block where the two halves of pass 2 meet, before its factorisation
  int *bad;     // [g][64]
  int ndg, dgw, nsg, K;  // dim groups per utterance, dims per group, system groups, chunks per utterance (from Tmax)
  int nw, mw;
  int narrow[kMaxNw];       // window w has l = u = 0
  int wspec;                // the usual set: three windows, the first one narrow, the others not
  double cpad[kMaxNw][5];   // coefficients, zero padded to [-EXT, EXT] (index j + EXT)
};

// ---- buffer loads / stores: wave-uniform descriptor, row offset in an SGPR, this lane's byte offset in one VGPR (a plain
// pointer per lane costs two address registers and a 64-bit add per access: this kernel has neither to spare) ----
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
template <typename T>
__device__ __forceinline__ T ld_buf(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
template <>
__device__ __forceinline__ double ld_buf<double>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 0);
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <>
__device__ __forceinline__ float ld_buf<float>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 0));
}
__device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const u32x2 w = {(unsigned)u, (unsigned)(u >> 32)};
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, loff, soff, 0);
}
__device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, loff, soff, 0);
}

__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}
// 1 / var: float32 inputs keep the reference's float32 reciprocal exactly (_mlpg.py:188); float64 to ~1 ulp (hardware seed + two
// Newton steps, as the strip and wave kernels: a correctly rounded division is 25 instructions, three per frame)
template <typename T>
__device__ __forceinline__ double tau_of(T v);
template <>
__device__ __forceinline__ double tau_of<float>(float v) { return recip_in_dtype<float>(v); }
template <>
__device__ __forceinline__ double tau_of<double>(double v) { return fast_rcp(v); }

// BWD (paramgen/_mlpg.py:202-281, mlpg_grad): the same matrix, the right-hand side is grad_out, and pass 3 ends with
// grad[t, w * sd + d] = tau_w[t] * sum_k c_w[l + k] z[t + k].  A row needs z up to EXT frames to its right, so a chunk writes the
// gradient rows f0 - EXT .. f0 + C - EXT - 1 (the previous separator's solution is at hand); the utterance's last chunk also writes
// its own last EXT rows.  The precisions are read again there (they were in the cache a few microseconds ago).
template <typename TIN, typename TOUT, int VM, int Q, bool P3, bool BWD = false>
__global__ __launch_bounds__(kW * 64, 2) void chunk_kernel(const Problem p, const Args a) {
  using G = Geo<Q>;
  constexpr int EXT = G::EXT, I = G::I, C = G::C, NLDS = P3 ? G::NLDS : 0, NF = C + 2 * EXT;
  extern __shared__ double lds_rows[];  // pass 3: [kW][NLDS][Q + 1][64]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long item = (long)blockIdx.x * kW + wv;
  if (item >= (long)a.nsg * a.K) return;
  const int g = (int)(item / a.K), c = (int)(item - (long)g * a.K);
  const int b = g / a.ndg, dg = g - b * a.ndg;
  const int Tmax = p.Tmax, sd = p.sd;
  int T = p.lengths ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  T = __builtin_amdgcn_readfirstlane(T);
  const int d0 = dg * a.dgw;
  const int nd = sd - d0 < a.dgw ? sd - d0 : a.dgw;
  const bool lane_ok = lane < nd;
  const int d = d0 + (lane_ok ? lane : nd - 1);  // idle lanes shadow the group's last dim (never stored)
  const int f0 = c * C;
  const __amdgpu_buffer_rsrc_t ors = make_rsrc((TOUT *)p.out + (size_t)b * Tmax * p.ld_out + d0);
  const unsigned ooff = (unsigned)(d - d0) * (unsigned)sizeof(TOUT), ldo_bytes = (unsigned)p.ld_out * (unsigned)sizeof(TOUT);
  const unsigned owin_bytes = (unsigned)sd * (unsigned)sizeof(TOUT);
  if (f0 >= T) {  // nothing of the utterance in this chunk
    if (P3 && lane_ok)
      for (int r = 0; r < C && f0 + r < Tmax; ++r)
        for (int w = 0; w < (BWD ? a.nw : 1); ++w) st_buf(ors, (unsigned)(f0 + r) * ldo_bytes + (unsigned)w * owin_bytes, ooff, (TOUT)0);
    return;
  }
  // forward: the means (window-major columns, stride ld_in); backward: grad_out (one column per dim, stride ld_gout)
  const __amdgpu_buffer_rsrc_t mrs = make_rsrc(BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * p.ld_gout + d0
                                                   : (const TIN *)p.mean + (size_t)b * Tmax * p.ld_in + d0);
  const unsigned ldg_bytes = (unsigned)p.ld_gout * (unsigned)sizeof(TIN);
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc(VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)b * Tmax * p.ld_in + d0
                                                                      : (VM == MLPG_HIP_VAR_GLOBAL ? (const TIN *)p.var + d0 : (const TIN *)p.out));
  const unsigned loff = (unsigned)(d - d0) * (unsigned)sizeof(TIN);
  const unsigned ld_bytes = (unsigned)p.ld_in * (unsigned)sizeof(TIN), win_bytes = (unsigned)sd * (unsigned)sizeof(TIN);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(a.rec + ((size_t)g * a.K + c) * G::kRec * 64);  // this chunk's record
  const unsigned roff = (unsigned)lane * 8u;
  const int nw = a.nw, mw = a.mw;
  double tau_g[kMaxNw];
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w)
    tau_g[w] = (VM == MLPG_HIP_VAR_GLOBAL && w < nw) ? tau_of<TIN>(ld_buf<TIN>(vrs, (unsigned)w * win_bytes, loff)) : 1.0;
  const double first = c == 0 ? 0.0 : 1.0;  // chunk 0: columns left of frame 0 do not exist

  // pass 3: the neighbouring separators' solutions
  double xl[Q], xr[Q];
  if (P3) {
    const double *xs = a.xs + ((size_t)g * a.K + c) * Q * 64 + lane;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      xr[q] = xs[q * 64];
      xl[q] = c ? xs[q * 64 - (long)Q * 64] : 0.0;
    }
  }

  // ---- the frames, requested PF ahead of their use ----
  constexpr int PF = 4;  // frames requested ahead of their use (12 registers each in float64; 2 ... 7: no difference, profiles/r04_notes.md section 9)
  TIN rv[NF][kMaxNw], rm[NF][kMaxNw];
  double acc[C][Q + 1], accb[C];  // rows of P and b under assembly: row r is open from frame slot s = r to s = r + 2 EXT
  double Lm[C][Q + 1];  // Lm[r][m]: multiplier of row r for column r - m (an interior pivot), else 0
  double dinv[C], ub[C];
  // pass 1: the left-coupling columns, forward-substituted
  double us_reg[C][Q];  // (in LDS instead -- a ring of 8 rows -- measured no faster once the sink problem below was understood)
  auto us_at = [&](const int r, const int q) __attribute__((always_inline)) -> double & { return us_reg[r][q]; };
  double sll[G::NS], gl[Q];
#pragma unroll
  for (int k = 0; k < G::NS; ++k) sll[k] = 0.0;
#pragma unroll
  for (int k = 0; k < Q; ++k) gl[k] = 0.0;
  bool bad = false;
  const int w_base = wv * (NLDS ? NLDS : 1) * (Q + 1) * 64 + lane;
  auto lds_w = [&](const int idx) __attribute__((always_inline)) -> double & { return lds_rows[w_base + idx * 64]; };

  // The chunk's loop, twice: FAST for a chunk whose frames (halo included) all lie inside the utterance, away from its first
  // and last mw frames, with the usual window set (static, then two dynamic windows) -- no condition left in it; the
  // general form for the chunks at an utterance's ends and for other window sets.  (An instruction of a 64-wide wavefront
  // issues over 4 cycles whatever it does: the masks, selects and register moves of the general form cost as much as its
  // arithmetic.)
  // (Chunk length: with more than 16 interior frames the same code, unrolled, needs more registers than a wavefront has.)
  auto run = [&](auto fast_c) __attribute__((always_inline)) {
  constexpr bool FAST = decltype(fast_c)::value;
  auto load_frame = [&](const int s) __attribute__((always_inline)) {  // s: frame f0 + s - EXT, slot s
    const int t = f0 + s - EXT;
    const bool fl = FAST || (t >= 0 && t < T);
#pragma unroll
    for (int w = 0; w < kMaxNw; ++w) {
      rv[s][w] = (TIN)1;
      rm[s][w] = (TIN)0;
      if (fl && (FAST || w < nw)) {
        if (VM == MLPG_HIP_VAR_FRAME) rv[s][w] = ld_buf<TIN>(vrs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, loff);
        if (!BWD) rm[s][w] = ld_buf<TIN>(mrs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, loff);
        else if (w == 0) rm[s][0] = ld_buf<TIN>(mrs, (unsigned)t * ldg_bytes, loff);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load_frame(s);
#pragma unroll
  for (int s = 0; s < NF; ++s) {
    if (s + PF < NF) load_frame(s + PF);
    // ---- frame t = f0 + s - EXT spreads its precisions over rows t - EXT .. t + EXT ----
    const int t = f0 + s - EXT;
    const bool fl = FAST || (t >= 0 && t < T);
    if (BWD && s - EXT >= 0 && s - EXT < C) accb[s - EXT] = fl ? (double)rm[s][0] : 0.0;  // backward: row r's right-hand side is grad_out[f0 + r]
    if (s < C) {  // the row this frame is the first to touch
      if (!BWD) accb[s] = 0.0;
#pragma unroll
      for (int k = 0; k <= Q; ++k) acc[s][k] = 0.0;
    }
#pragma unroll
    for (int w = 0; w < kMaxNw; ++w) {
      if (FAST || w < nw) {
        const bool lv = FAST || (fl && (w == 0 || (mw != 0 && t >= mw && t < T - mw)));
        double tau = 0.0;
        if (lv) tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(rv[s][w]) : (VM == MLPG_HIP_VAR_GLOBAL ? tau_g[w] : 1.0);
        // tau * c[a] per lane, times c[b] from a scalar register: 5 scalar constants per window.  (With the 15 products
        // c[a] c[b] per window as kernel arguments the three windows need 120 scalar registers; the compiler re-read them
        // from the argument segment a dozen times per frame, an s_waitcnt each: 0.54 ms instead of 0.3x.)
        const double mu = (double)rm[s][w];
        if (FAST ? w == 0 : a.narrow[w] != 0) {
          const int r = s - EXT;
          if (r >= 0 && r < C) {
            const double t0 = tau * a.cpad[w][EXT];
            acc[r][0] = __builtin_fma(t0, a.cpad[w][EXT], acc[r][0]);
            if (!BWD) accb[r] = __builtin_fma(t0, mu, accb[r]);
          }
        } else {
          double ta[2 * EXT + 1];
#pragma unroll
          for (int j = 0; j <= 2 * EXT; ++j) ta[j] = tau * a.cpad[w][j];
#pragma unroll
          for (int j1 = -EXT; j1 <= EXT; ++j1) {
            const int r = s - EXT + j1;  // local row
            if (r < 0 || r >= C) continue;
            if (!BWD) accb[r] = __builtin_fma(ta[j1 + EXT], mu, accb[r]);
#pragma unroll
            for (int j2 = -EXT; j2 <= j1; ++j2) {
              const int r2 = s - EXT + j2;  // local column
              if (r2 < 0 && !FAST) acc[r][j1 - j2] = __builtin_fma(ta[j1 + EXT] * first, a.cpad[w][j2 + EXT], acc[r][j1 - j2]);
              else acc[r][j1 - j2] = __builtin_fma(ta[j1 + EXT], a.cpad[w][j2 + EXT], acc[r][j1 - j2]);
            }
          }
        }
      }
    }
    // ---- row r = s - 2 EXT is complete: eliminate it ----
    const int r = s - 2 * EXT;
    __builtin_amdgcn_sched_barrier(0);
    if (r < 0) continue;
    double A[Q + 1], yb = accb[r];
#pragma unroll
    for (int k = 0; k <= Q; ++k) A[k] = acc[r][k];
    if (!FAST && f0 + r >= T) {  // behind the utterance's end: an identity row
      yb = 0.0;
      A[0] = 1.0;
#pragma unroll
      for (int k = 1; k <= Q; ++k) A[k] = 0.0;
    }
    double ys[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) ys[q] = 0.0;
    double tt[Q + 1], row[Q + 1];
#pragma unroll
    for (int k = Q; k >= 1; --k) {
      tt[k] = row[k] = 0.0;
      if (r - k < 0) {  // a column of the previous separator: coupling, not a pivot
        if (P3) yb = __builtin_fma(-A[k], xl[Q + r - k], yb);
        else ys[Q + r - k] = A[k];
        continue;
      }
      double num = A[k];
#pragma unroll
      for (int m = k + 1; m <= Q; ++m)
        if (r - m >= 0) num = __builtin_fma(-tt[m], Lm[r - k][m - k], num);
      if (r - k < I) {
        tt[k] = num;
        row[k] = num * dinv[r - k];
      } else {
        // separator row against an earlier separator row: a Schur entry
        if (!P3) st_buf(rrs, (unsigned)(G::oSRR + tri(r - I, r - k - I)) * 512u, roff, num);
      }
    }
    double diag = A[0];
#pragma unroll
    for (int m = 1; m <= Q; ++m)
      if (r - m >= 0) diag = __builtin_fma(-tt[m], row[m], diag);
#pragma unroll
    for (int m = 1; m <= Q; ++m) {
      if (r - m < 0 || r - m >= I) continue;  // (separator rows are no pivots: their multiplier is zero, their us undefined)
      yb = __builtin_fma(-row[m], ub[r - m], yb);
      if (!P3) {
#pragma unroll
        for (int q = 0; q < Q; ++q) ys[q] = __builtin_fma(-row[m], us_at(r - m, q), ys[q]);
      }
    }
#pragma unroll
    for (int m = 0; m <= Q; ++m) Lm[r][m] = m ? row[m] : 0.0;
    if (r < I) {
      bad = bad || !(diag > 0.0);
      const double di = fast_rcp(diag);
      dinv[r] = di;
      if (P3) {
        ub[r] = yb;
      } else {
        ub[r] = yb;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          us_at(r, q) = ys[q];
          const double wq = ys[q] * di;
          gl[q] = __builtin_fma(-wq, yb, gl[q]);
#pragma unroll
          for (int q2 = 0; q2 <= q; ++q2) sll[tri(q, q2)] = __builtin_fma(-wq, ys[q2], sll[tri(q, q2)]);
        }
        // The sums are needed only after the last interior row, and the compiler knows it: left alone it sinks all 14 chains
        // down to that point and keeps every row's ys alive until then (190 spilled registers).  An opaque use per row pins them.
#pragma unroll
        for (int k = 0; k < G::NS; ++k) asm volatile("" : "+v"(sll[k]));
#pragma unroll
        for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(gl[q]));
      }
    } else {
      dinv[r] = 0.0;
      ub[r] = yb;
      if (!P3) {  // a separator row: its part of the record is final (later rows do not pivot on it)
        st_buf(rrs, (unsigned)(G::oSRR + tri(r - I, r - I)) * 512u, roff, diag);
        st_buf(rrs, (unsigned)(G::oGR + r - I) * 512u, roff, yb);
#pragma unroll
        for (int q = 0; q < Q; ++q) st_buf(rrs, (unsigned)(G::oSRL + (r - I) * Q + q) * 512u, roff, ys[q]);
      }
    }
    if (!P3 && r == I - 1) {  // the last interior row: what the chunk adds to the previous separator is final
#pragma unroll
      for (int k = 0; k < G::NS; ++k) st_buf(rrs, (unsigned)(G::oSLL + k) * 512u, roff, sll[k]);
#pragma unroll
      for (int q = 0; q < Q; ++q) st_buf(rrs, (unsigned)(G::oGL + q) * 512u, roff, gl[q]);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise hoists the later rows' loads and LDS reads over this one)
    if (P3 && NLDS) {  // the row that leaves the window of the forward pass goes to LDS
      const int ro = r - Q;
      if (ro >= 0 && ro < NLDS) {
#pragma unroll
        for (int m = 1; m <= Q; ++m) lds_w(ro * (Q + 1) + m) = Lm[ro][m];
        lds_w(ro * (Q + 1)) = ub[ro] * dinv[ro];
      }
    }
  }
  };  // run
  const bool fast = a.wspec && f0 - EXT >= mw && f0 + C + EXT <= T - mw;
  if (fast) run(std::integral_constant<bool, true>{});
  else run(std::integral_constant<bool, false>{});

  if (bad) a.bad[(size_t)g * 64 + lane] = 1;
  if (!P3) return;
  // ---- pass 3: back-substitution and output ----
  double x[C];
#pragma unroll
  for (int q = 0; q < Q; ++q) x[I + q] = xr[q];
#pragma unroll
  for (int r = I - 1; r >= 0; --r) {
    double v;
    if (r < NLDS) {
      v = lds_w(r * (Q + 1));
    } else {
      v = ub[r] * dinv[r];
    }
#pragma unroll
    for (int m = 1; m <= Q; ++m) {
      if (r + m >= C) continue;
      // Lm[r + m][m]: row r + m's multiplier for column r; rows below NLDS were handed to LDS
      const double l = (r + m < NLDS) ? lds_w((r + m) * (Q + 1) + m) : Lm[r + m][m];
      v = __builtin_fma(-l, x[r + m], v);
    }
    x[r] = v;
  }
  if (!BWD) {
    if (lane_ok) {
#pragma unroll
      for (int r = 0; r < C; ++r) {
        const int t = f0 + r;
        if (t < Tmax) st_buf(ors, (unsigned)t * ldo_bytes, ooff, t < T ? (TOUT)x[r] : (TOUT)0);
      }
    }
    return;
  }
  // ---- backward epilogue: grad[t, w] = tau_w[t] * sum_k c_w[l + k] z[t + k] for the rows this chunk is responsible for ----
  // z at local index i (frame f0 + i), i in [-Q, C + EXT): the previous separator, this chunk, nothing beyond the utterance
  const bool lastc = f0 + C >= T;  // the utterance ends in this chunk: it writes its last EXT rows itself
  auto z_at = [&](const int i) __attribute__((always_inline)) -> double {
    if (i < 0) return i >= -Q ? xl[Q + i] : 0.0;
    if (i < C) return (f0 + i < T) ? x[i] : 0.0;
    return 0.0;  // (only read by the last chunk's own rows: frames behind the end)
  };
#pragma unroll
  for (int i = -EXT; i < C; ++i) {
    const int t = f0 + i;
    const bool mine = i < C - EXT || lastc;
    if (t < 0 || t >= Tmax || !mine) continue;
    const bool fl = t < T;
#pragma unroll
    for (int w = 0; w < kMaxNw; ++w) {
      if (w >= nw) continue;
      double gval = 0.0;
      const bool lv = fl && (w == 0 || (mw != 0 && t >= mw && t < T - mw));
      if (lv) {
        double tau = 1.0;
        if (VM == MLPG_HIP_VAR_FRAME) tau = tau_of<TIN>(ld_buf<TIN>(vrs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, loff));
        else if (VM == MLPG_HIP_VAR_GLOBAL) tau = tau_g[w];
        double acc_w = 0.0;
#pragma unroll
        for (int j = -EXT; j <= EXT; ++j) acc_w = __builtin_fma(a.cpad[w][j + EXT], z_at(i + j), acc_w);
        gval = tau * acc_w;
      }
      if (lane_ok) st_buf(ors, (unsigned)t * ldo_bytes + (unsigned)w * owin_bytes, ooff, (TOUT)gval);
    }
  }
}

// ---- pass 2: the separators' block-tridiagonal system ----
// Two wavefronts per (utterance, dim group) that never talk to each other inside a kernel: reduce_elim_kernel eliminates the first
// half of the blocks downwards (h = 0) and the second half upwards (h = 1), reduce_subst_kernel joins the two at the middle
// (both wavefronts solve the same two-block system) and substitutes outwards.  One wavefront walking all the blocks is issue-bound
// on its own instruction stream (~1 us per block, 120 us at T = 1000); the halves take half of that.
template <int Q>
struct Blk {
  using G = Geo<Q>;
  static constexpr int NS = G::NS;
  // solve (L D L^T) v = e with the factor held as (Lf strict lower, di inverse pivots)
  static __device__ __forceinline__ void solve(const double (&Lf)[NS], const double (&di)[Q], double (&v)[Q]) {
#pragma unroll
    for (int i = 1; i < Q; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) v[i] = __builtin_fma(-Lf[tri(i, j)], v[j], v[i]);
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] *= di[i];
#pragma unroll
    for (int i = Q - 2; i >= 0; --i)
#pragma unroll
      for (int j = i + 1; j < Q; ++j) v[i] = __builtin_fma(-Lf[tri(j, i)], v[j], v[i]);
  }
  // D (packed symmetric) -> unit-lower multipliers in its strict lower part, inverse pivots in di; false on a non-positive pivot
  static __device__ __forceinline__ bool factor(double (&D)[NS], double (&di)[Q]) {
    bool ok = true;
    double dv[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      double piv = D[tri(j, j)];
#pragma unroll
      for (int m = 0; m < j; ++m) piv = __builtin_fma(-D[tri(j, m)] * D[tri(j, m)], dv[m], piv);
      ok = ok && (piv > 0.0);
      dv[j] = piv;
      di[j] = fast_rcp(piv);
#pragma unroll
      for (int i = j + 1; i < Q; ++i) {
        double v = D[tri(i, j)];
#pragma unroll
        for (int m = 0; m < j; ++m) v = __builtin_fma(-D[tri(i, m)] * D[tri(j, m)], dv[m], v);
        D[tri(i, j)] = v * di[j];
      }
    }
    return ok;
  }
  // (D, r) -= coupling through the neighbouring block (factor Lp, dpi; right-hand side rp).  DOWN: the neighbour is the block
  // above, E[q][s] couples this block's row q to its column s;  !DOWN: the neighbour is the block below, E[q][s] couples ITS row
  // q to this block's column s.
  template <bool DOWN>
  static __device__ __forceinline__ void couple(double (&D)[NS], double (&r)[Q], const double (&E)[Q * Q], const double (&Lp)[NS],
                                                const double (&dpi)[Q], const double (&rp)[Q]) {
#pragma unroll
    for (int i = 0; i < Q; ++i) {  // row i of E Dp^-1 (DOWN) / E^T Dp^-1
      double gq[Q];
#pragma unroll
      for (int s = 0; s < Q; ++s) gq[s] = DOWN ? E[i * Q + s] : E[s * Q + i];
      solve(Lp, dpi, gq);
#pragma unroll
      for (int s = 0; s < Q; ++s) {
        r[i] = __builtin_fma(-gq[s], rp[s], r[i]);
#pragma unroll
        for (int i2 = 0; i2 <= i; ++i2) D[tri(i, i2)] = __builtin_fma(-gq[s], DOWN ? E[i2 * Q + s] : E[s * Q + i2], D[tri(i, i2)]);
      }
    }
  }
  static __device__ __forceinline__ void store_fac(double *fk, const double (&D)[NS], const double (&di)[Q], const double (&r)[Q]) {
    int o = 0;
#pragma unroll
    for (int i = 1; i < Q; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) fk[(o++) * 64] = D[tri(i, j)];
#pragma unroll
    for (int i = 0; i < Q; ++i) fk[(NS - Q + i) * 64] = di[i];
#pragma unroll
    for (int i = 0; i < Q; ++i) fk[(NS + i) * 64] = r[i];
  }
  static __device__ __forceinline__ void unpack_fac(const double (&pF)[G::kFac], double (&Lf)[NS], double (&di)[Q], double (&r)[Q]) {
    int o = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) Lf[i] = 0.0;
#pragma unroll
    for (int i = 1; i < Q; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) Lf[tri(i, j)] = pF[o++];
#pragma unroll
    for (int i = 0; i < Q; ++i) di[i] = pF[NS - Q + i];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = pF[NS + i];
  }
};

// mid: [g][kFac][64] -- the middle block of the upward half BEFORE it is factorised (packed D, then r)
template <int Q>
__global__ __launch_bounds__(64) void reduce_elim_kernel(const Problem p, const Args a) {
  using G = Geo<Q>;
  using B = Blk<Q>;
  constexpr int NS = G::NS, C = G::C;
  const int lane = threadIdx.x;
  const int g = blockIdx.x >> 1, h = blockIdx.x & 1;
  const int b = g / a.ndg;
  int T = p.lengths ? p.lengths[b] : p.Tmax;
  T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
  const int Kb = (T + C - 1) / C;  // chunks that hold frames of this utterance
  const int m = Kb / 2;            // blocks 0 .. m-1 are eliminated downwards, m .. Kb-1 upwards
  const int k0 = h ? Kb - 1 : 0, k1 = h ? m - 1 : m, dk = h ? -1 : 1;  // k0, k0 + dk, ... while != k1
  if (k0 == k1) return;
  const double *rec = a.rec + (size_t)g * a.K * G::kRec * 64 + lane;
  double *fac = a.fac + (size_t)g * a.K * G::kFac * 64 + lane;
  bool bad = false;
  double Lp[NS], dpi[Q], rp[Q];
  // block k: diagonal (two parts), right-hand side (two parts), coupling to the block eliminated before it -- requested kPF2
  // blocks ahead (a block is 22 KB per wavefront and a step is shorter than the memory's latency)
  constexpr int kPF2 = 3, kBlk = 2 * NS + 2 * Q + Q * Q;
  double pb[kPF2][kBlk];
  auto load = [&](double (&dst)[kBlk], const int k) __attribute__((always_inline)) {
    const bool in = h ? k > k1 : k < k1;
    const double *rk = rec + (size_t)(in ? k : k0) * G::kRec * 64;
    const bool nx = in && k + 1 < Kb;
    const double *rn = rk + (nx ? (size_t)G::kRec * 64 : 0);
    const int ke = h ? k + 1 : k;  // downwards: E_k (record k couples separator k to k-1); upwards: E_{k+1}
    const bool he = in && ke >= 1 && ke < Kb;
    const double *re = rec + (size_t)(he ? ke : k0) * G::kRec * 64;
#pragma unroll
    for (int i = 0; i < NS; ++i) dst[i] = rk[(G::oSRR + i) * 64];
#pragma unroll
    for (int i = 0; i < Q; ++i) dst[NS + i] = rk[(G::oGR + i) * 64];
#pragma unroll
    for (int i = 0; i < NS; ++i) dst[NS + Q + i] = nx ? rn[(G::oSLL + i) * 64] : 0.0;
#pragma unroll
    for (int i = 0; i < Q; ++i) dst[2 * NS + Q + i] = nx ? rn[(G::oGL + i) * 64] : 0.0;
#pragma unroll
    for (int i = 0; i < Q * Q; ++i) dst[2 * NS + 2 * Q + i] = he ? re[(G::oSRL + i) * 64] : 0.0;
  };
#pragma unroll
  for (int j = 0; j < kPF2; ++j) load(pb[j], k0 + j * dk);
  // One step: block k sits in ring slot `slot` (a compile-time index: the k loop is unrolled by kPF2; until round 5 the slots
  // were rotated with register moves).  Measured neutral (72.9 vs 67 us for 256 x 50 blocks): the pass is bound by READING the
  // records -- 270 MB at the config-2 shape, 4 TB/s from 512 wavefronts with three blocks each in flight -- not by latency.
  auto step = [&](auto slot_c, const int k) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_c)::value;
    double D[NS], r[Q], E[Q * Q];
#pragma unroll
    for (int i = 0; i < NS; ++i) D[i] = pb[slot][i] + pb[slot][NS + Q + i];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = pb[slot][NS + i] + pb[slot][2 * NS + Q + i];
#pragma unroll
    for (int i = 0; i < Q * Q; ++i) E[i] = pb[slot][2 * NS + 2 * Q + i];
    load(pb[slot], k + kPF2 * dk);
    if (k != k0) {
      if (h) B::template couple<false>(D, r, E, Lp, dpi, rp);
      else B::template couple<true>(D, r, E, Lp, dpi, rp);
    }
    if (h && k == m) {  // the block where the halves meet: kept unfactorised for reduce_subst_kernel
      double *mk = a.mid + (size_t)g * G::kFac * 64 + lane;
#pragma unroll
      for (int i = 0; i < NS; ++i) mk[i * 64] = D[i];
#pragma unroll
      for (int i = 0; i < Q; ++i) mk[(NS + i) * 64] = r[i];
    }
    double di[Q];
    bad = !B::factor(D, di) || bad;
    B::store_fac(fac + (size_t)k * G::kFac * 64, D, di, r);
#pragma unroll
    for (int i = 0; i < NS; ++i) Lp[i] = D[i];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      dpi[i] = di[i];
      rp[i] = r[i];
    }
  };
  static_assert(kPF2 == 3, "the k loop below is unrolled by three");
  for (int k = k0;;) {
    if (k == k1) break;
    step(std::integral_constant<int, 0>{}, k); k += dk;
    if (k == k1) break;
    step(std::integral_constant<int, 1>{}, k); k += dk;
    if (k == k1) break;
    step(std::integral_constant<int, 2>{}, k); k += dk;
  }
  if (bad) a.bad[(size_t)g * 64 + lane] = 1;
}

template <int Q>
__global__ __launch_bounds__(64) void reduce_subst_kernel(const Problem p, const Args a) {
  using G = Geo<Q>;
  using B = Blk<Q>;
  constexpr int NS = G::NS, C = G::C;
  const int lane = threadIdx.x;
  const int g = blockIdx.x >> 1, h = blockIdx.x & 1;
  const int b = g / a.ndg;
  int T = p.lengths ? p.lengths[b] : p.Tmax;
  T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
  const int Kb = (T + C - 1) / C;
  if (Kb == 0) return;
  const int m = Kb / 2;
  if (h == 0 && m == 0) return;
  const double *rec = a.rec + (size_t)g * a.K * G::kRec * 64 + lane;
  const double *fac = a.fac + (size_t)g * a.K * G::kFac * 64 + lane;
  double *xs = a.xs + (size_t)g * a.K * Q * 64 + lane;
  bool bad = false;
  // ---- the middle: x_m (and, downwards, x_{m-1}) ----
  double xm[Q];
  {
    double pF[G::kFac], D[NS], r[Q];
    const double *mk = a.mid + (size_t)g * G::kFac * 64 + lane;
#pragma unroll
    for (int i = 0; i < G::kFac; ++i) pF[i] = mk[i * 64];
#pragma unroll
    for (int i = 0; i < NS; ++i) D[i] = pF[i];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = pF[NS + i];
    if (m >= 1) {
      double pG[G::kFac], E[Q * Q], Lp[NS], dpi[Q], rp[Q];
      const double *fk = fac + (size_t)(m - 1) * G::kFac * 64;
#pragma unroll
      for (int i = 0; i < G::kFac; ++i) pG[i] = fk[i * 64];
      const double *re = rec + (size_t)m * G::kRec * 64;
#pragma unroll
      for (int i = 0; i < Q * Q; ++i) E[i] = re[(G::oSRL + i) * 64];
      B::unpack_fac(pG, Lp, dpi, rp);
      B::template couple<true>(D, r, E, Lp, dpi, rp);
    }
    double di[Q];
    bad = !B::factor(D, di);
#pragma unroll
    for (int i = 0; i < Q; ++i) xm[i] = r[i];
    B::solve(D, di, xm);
  }
  if (h) {
    // upwards from the middle: x_k = D''_k^-1 (r''_k - E_k x_{k-1})
#pragma unroll
    for (int i = 0; i < Q; ++i) xs[((size_t)m * Q + i) * 64] = xm[i];
    constexpr int kPF2 = 3;
    double pF[kPF2][G::kFac], pE[kPF2][Q * Q];
    auto load_up = [&](const int j, const int k) __attribute__((always_inline)) {
      const int kk = k < Kb ? k : Kb - 1;
      const double *fk = fac + (size_t)kk * G::kFac * 64;
#pragma unroll
      for (int i = 0; i < G::kFac; ++i) pF[j][i] = fk[i * 64];
      const double *re = rec + (size_t)kk * G::kRec * 64;
#pragma unroll
      for (int i = 0; i < Q * Q; ++i) pE[j][i] = re[(G::oSRL + i) * 64];
    };
#pragma unroll
    for (int j = 0; j < kPF2; ++j) load_up(j, m + 1 + j);
    auto step_up = [&](auto slot_c, const int k) __attribute__((always_inline)) {   // (static ring slots: see reduce_elim_kernel)
      constexpr int slot = decltype(slot_c)::value;
      double Lf[NS], di[Q], v[Q], E[Q * Q];
      B::unpack_fac(pF[slot], Lf, di, v);
#pragma unroll
      for (int i = 0; i < Q * Q; ++i) E[i] = pE[slot][i];
      load_up(slot, k + kPF2);
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int s = 0; s < Q; ++s) v[q] = __builtin_fma(-E[q * Q + s], xm[s], v[q]);
      B::solve(Lf, di, v);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        xm[i] = v[i];
        xs[((size_t)k * Q + i) * 64] = v[i];
      }
    };
    static_assert(kPF2 == 3, "the k loop below is unrolled by three");
    for (int k = m + 1;;) {
      if (k >= Kb) break;
      step_up(std::integral_constant<int, 0>{}, k); ++k;
      if (k >= Kb) break;
      step_up(std::integral_constant<int, 1>{}, k); ++k;
      if (k >= Kb) break;
      step_up(std::integral_constant<int, 2>{}, k); ++k;
    }
  } else {
    // downwards from the middle: x_k = D'_k^-1 (r'_k - E_{k+1}^T x_{k+1}), k = m-1 .. 0
    constexpr int kPF2 = 3;
    double pF[kPF2][G::kFac], pE[kPF2][Q * Q];
    auto load_dn = [&](const int j, const int k) __attribute__((always_inline)) {
      const int kk = k >= 0 ? k : 0;
      const double *fk = fac + (size_t)kk * G::kFac * 64;
#pragma unroll
      for (int i = 0; i < G::kFac; ++i) pF[j][i] = fk[i * 64];
      const double *rn = rec + (size_t)(kk + 1) * G::kRec * 64;
#pragma unroll
      for (int i = 0; i < Q * Q; ++i) pE[j][i] = rn[(G::oSRL + i) * 64];
    };
#pragma unroll
    for (int j = 0; j < kPF2; ++j) load_dn(j, m - 1 - j);
    auto step_dn = [&](auto slot_c, const int k) __attribute__((always_inline)) {
      constexpr int slot = decltype(slot_c)::value;
      double Lf[NS], di[Q], v[Q], E[Q * Q];
      B::unpack_fac(pF[slot], Lf, di, v);
#pragma unroll
      for (int i = 0; i < Q * Q; ++i) E[i] = pE[slot][i];
      load_dn(slot, k - kPF2);
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int s = 0; s < Q; ++s) v[s] = __builtin_fma(-E[q * Q + s], xm[q], v[s]);
      B::solve(Lf, di, v);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        xm[i] = v[i];
        xs[((size_t)k * Q + i) * 64] = v[i];
      }
    };
    static_assert(kPF2 == 3, "the k loop below is unrolled by three");
    for (int k = m - 1;;) {
      if (k < 0) break;
      step_dn(std::integral_constant<int, 0>{}, k); --k;
      if (k < 0) break;
      step_dn(std::integral_constant<int, 1>{}, k); --k;
      if (k < 0) break;
      step_dn(std::integral_constant<int, 2>{}, k); --k;
    }
  }
  if (bad) a.bad[(size_t)g * 64 + lane] = 1;
}

// One thread per system, after pass 3: marked systems get the reference's verdict (natural-order first failing pivot;
// -2 if that scan finds none: the blocked elimination broke down on a numerically singular system) and a zero column.
template <typename TIN, typename TOUT, int Q, bool BWD = false>
__global__ __launch_bounds__(256) void verdict_kernel(const Problem p, const WinSet ws, const Args a) {
  const long s = (long)blockIdx.x * 256 + threadIdx.x;
  if (s >= (long)p.B * p.sd) return;
  const int b = (int)(s / p.sd), d = (int)(s - (long)b * p.sd);
  const int dg = d / a.dgw, lane = d - dg * a.dgw;
  int *flag = a.bad + ((size_t)b * a.ndg + dg) * 64 + lane;
  int status = 0;
  if (*flag) {
    *flag = 0;  // (the next launch on this stream finds the marks cleared)
    int T = p.lengths ? p.lengths[b] : p.Tmax;
    T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
    const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
    status = first_bad_pivot<Q, TIN, BWD>(view, ws);
    if (status == 0) status = -2;
    TOUT *out_b = (TOUT *)p.out + (size_t)b * p.Tmax * p.ld_out + d;
    for (int t = 0; t < p.Tmax; ++t)
      for (int w = 0; w < (BWD ? ws.nw : 1); ++w) out_b[(size_t)t * p.ld_out + (size_t)w * p.sd] = (TOUT)0;
  }
  if (p.status) p.status[(size_t)b * p.ld_status + d] = status;
}


// ---- host side: one launch sequence per (dtype, direction), instantiated in mlpg_chunk_{fwd,bwd}_{f64,f32}.hip (one translation unit
// held all of them until round 6: 89 s of a 3-minute cold build on its own) ----
template <int Q>
void fill_args(const Problem &p, const WinSet &ws, Args *a) {
  constexpr int EXT = Geo<Q>::EXT;
  a->nw = ws.nw;
  a->mw = ws.mw;
  a->wspec = (ws.nw == 3 && ws.l[0] == 0 && ws.u[0] == 0 && (ws.l[1] | ws.u[1]) && (ws.l[2] | ws.u[2])) ? 1 : 0;
  for (int w = 0; w < kMaxNw; ++w) {
    a->narrow[w] = 0;
    for (int j = 0; j < 5; ++j) a->cpad[w][j] = 0.0;
    if (w >= ws.nw) continue;
    const int l = ws.l[w], u = ws.u[w];
    a->narrow[w] = (l == 0 && u == 0) ? 1 : 0;
    // W_w[t, t + j] = c_w[l + j], j in [-l, u]
    for (int j = -l; j <= u; ++j) a->cpad[w][j + EXT] = ws.c[ws.off[w] + l + j];
  }
}

// Measured and dropped (tools/gpurun/r4_chunk.sh history, profiles/r04_notes.md): slabs of utterances sized to stay in the Infinity
// Cache between pass 1 and pass 3 (48 / 96 / 192 MB: 2.10 / 1.15 / 0.84 ms against 0.58 for the whole batch: pass 2's ~120 us of
// sequential latency is paid per slab); pass 2 of one part of the batch on a side stream under pass 1 / pass 3 of the others
// (2 / 3 / 4 parts: 0.70 / 0.75 / 0.76 ms against 0.54: the chunk kernels book the whole register file, the side stream's
// wavefronts wait for them to drain).
template <typename TIN, typename TOUT, int Q, bool BWD>
int launch_q(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  using G = Geo<Q>;
  Args a;
  fill_args<Q>(p, ws, &a);
  a.dgw = p.sd < 64 ? p.sd : 64;
  a.ndg = (p.sd + a.dgw - 1) / a.dgw;
  a.K = (p.Tmax + G::C - 1) / G::C;
  a.nsg = p.B * a.ndg;
  const size_t nsg = (size_t)a.nsg;
  const size_t rec_b = nsg * a.K * G::kRec * 64 * sizeof(double), fac_b = nsg * a.K * G::kFac * 64 * sizeof(double),
               xs_b = nsg * a.K * Q * 64 * sizeof(double), mid_b = nsg * G::kFac * 64 * sizeof(double), bad_b = nsg * 64 * sizeof(int);
  char *sc = (char *)scratch(device, st, 6, rec_b + fac_b + xs_b + mid_b + bad_b + 256);
  if (!sc) return MLPG_HIP_ENOMEM;
  a.rec = (double *)sc;
  a.fac = (double *)(sc + rec_b);
  a.xs = (double *)(sc + rec_b + fac_b);
  a.mid = (double *)(sc + rec_b + fac_b + xs_b);
  a.bad = (int *)(sc + rec_b + fac_b + xs_b + mid_b);
  MLPG_HIP_CHECK(hipMemsetAsync(a.bad, 0, bad_b, st));  // the marks of non-positive pivots
  constexpr size_t lds3 = (size_t)kW * (G::NLDS ? G::NLDS : 0) * (Q + 1) * 64 * sizeof(double);
  constexpr size_t lds1 = 0;
  const long items = (long)a.nsg * a.K;
  const dim3 grid((unsigned)((items + kW - 1) / kW)), block(kW * 64);
  note_launch(kCountChunk);
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_FRAME, Q, false, BWD>), grid, block, lds1, st, p, a); break;
    case MLPG_HIP_VAR_GLOBAL: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_GLOBAL, Q, false, BWD>), grid, block, lds1, st, p, a); break;
    default: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_UNIT, Q, false, BWD>), grid, block, lds1, st, p, a);
  }
  hipLaunchKernelGGL((reduce_elim_kernel<Q>), dim3((unsigned)(2 * a.nsg)), dim3(64), 0, st, p, a);
  hipLaunchKernelGGL((reduce_subst_kernel<Q>), dim3((unsigned)(2 * a.nsg)), dim3(64), 0, st, p, a);
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_FRAME, Q, true, BWD>), grid, block, lds3, st, p, a); break;
    case MLPG_HIP_VAR_GLOBAL: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_GLOBAL, Q, true, BWD>), grid, block, lds3, st, p, a); break;
    default: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_UNIT, Q, true, BWD>), grid, block, lds3, st, p, a);
  }
  hipLaunchKernelGGL((verdict_kernel<TIN, TOUT, Q, BWD>), dim3((unsigned)(((long)p.B * p.sd + 255) / 256)), dim3(256), 0, st, p, ws, a);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  return ws.mw == 2 ? launch_q<TIN, TOUT, 4, BWD>(st, p, ws, device) : launch_q<TIN, TOUT, 2, BWD>(st, p, ws, device);
}


}  // namespace chunk
}  // namespace mlpg
