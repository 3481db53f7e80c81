// Unit-variance MLPG on float32 tensors as a FIR filter (MLPG_HIP_ALGO_FIR): autograd.unit_variance_mlpg forward / backward,
// BASELINE config 3.  Executable specification: tools/fir_model.py.
//
// With unit variances P = sum_w W~_w^T W_w is one matrix for every system of the launch (paramgen/_mlpg.py:297-373); the reference
// multiplies by the dense float32 R = P^-1 [W~_w^T] (autograd/_impl/mlpg.py:108-172).  Away from an utterance's ends P is Toeplitz and
// its inverse decays by about a bit per frame, so to 2^-26 of the largest tap
//     y = P^-1 b,   b[i] = sum_w sum_t W_w[t, i] m_w[t] mu_w[t]
// is a FIR filter of 2 H + 1 = 49 taps on b in the interior plus a table of E = 24 rows at either end (the matrix near the end does not
// depend on T).  No recurrence, no chain: lane = static dim, wavefront = a tile of 32 frames, every tile independent -- what a launch of
// 64 utterances needs to fill 256 CUs.  The table (1 + 2 E rows of 49 taps) comes from ONE exact solve of 49 one-hot right-hand sides on
// a 160-frame system with the library's own kernels, once per (device, window set); the decay is CHECKED there, a window set that does
// not decay to 2^-26 within 24 taps is refused and the other kernels take the call.
// Backward (autograd/_impl/mlpg.py:145-172: R^T grad): z = P^-1 grad_out by the same table (P^-1 is symmetric), then
// grad[t, w] = m_w[t] sum_k c_w[l + k] z[t + k].
// Float32 in, float32 out, float32 arithmetic (the reference's is a float32 GEMM over 3 T terms; here 49 terms per row, two rows per
// v_pk_fma_f32).  One launch: the first 2 nsg workgroups take the ends, the others eight tiles each (fir_kernel at the end of the device
// code).  The MSE instances are the training step mlpg_hip_unit_mse_step in this form (launch_fir_mse): two launches.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include <type_traits>

#include "common.h"

namespace mlpg {

int dispatch_solve(hipStream_t st, int in_dtype, int out_dtype, int algo, bool backward, const Problem &p, const WinSet &ws, int device);

namespace fir {

constexpr int kH = 24, kE = 24, kTaps = 2 * kH + 1, kRows = 1 + 2 * kE, kTT = 32, kEXT = 2, kTref = 160, kW = 8, kMaxNw = 3;

struct Args {
  const float *tap;  // [kRows][kTaps]: row 0 interior, 1..E rows t = 0..E-1, E+1..2E rows T-1 .. T-E with their taps in reverse order
  int ndg, dgw, nsg, nt;  // dim groups per utterance, dims per group, (utterance, dim group) pairs, tiles per utterance
  int nw, mw;
  float cpad[kMaxNw][2 * kEXT + 1];  // window coefficients, zero padded to [-2, 2]
  // the training step (MSE instances): forward writes dy = scale (y - target) and one partial sum of (y - target)^2 per wavefront,
  // backward reads dy; its last workgroup adds the partials up in a fixed order
  const float *target;  // (B, T, sd)
  float *dy;            // (B, T, sd)
  double *partials;     // one per wavefront of the forward launch
  double *loss;
  float scale;          // 2 / n_elems
  double inv_n;         // 1 / n_elems
  int npart;
  int y_given;          // forward: p.out holds y_out (else the trajectory is not stored)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
// Raw buffer access: address = base + soff (scalar: the frame's row) + loff (per lane: the dim).  The hardware checks loff -- not soff --
// against the descriptor's 2^31 - 1 bytes; the kernels use that as their mask: with loff = 0x80000000 a load returns 0 and a store is
// dropped, without a branch and without a select on the data (rows_fit_buffer keeps every real offset below 2^31).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float ld_f32(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 0));
}
__device__ __forceinline__ void st_f32(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, loff, soff, 0);
}

__device__ __forceinline__ double wave_sum(double v) {  // lanes added in a fixed order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// fir_tiles: one wavefront = one tile of kTT frames of one (utterance, dim group), every row with the INTERIOR taps.
// Written for a wavefront that is alone on its SIMD (64 utterances x 16 tiles = one wavefront per SIMD of the chip):
//   - branch-free: a frame outside the utterance is read from the nearest frame inside and enters with coefficient 0 (the
//     coefficients are scalars selected per frame); a row that is not this kernel's is stored to an offset the buffer drops;
//   - the 252 loads of a forward tile are issued in batches, each a batch ahead of its use (left to itself the compiler keeps 4-6 in
//     flight and the tile costs 84 round trips);
//   - the taps sit in the lanes of one register and are read out one at a time (v_readlane): 49 scalars held live cost more scalar
//     registers than there are, and the reloads from the kernel arguments that followed cost more than the arithmetic.
template <bool BWD, int EXT, bool MSE>
__device__ __forceinline__ void fir_tiles(const Problem &p, const Args &a, const unsigned blk) {
  constexpr int H = kH, E = kE, TT = kTT;
  // FIR outputs of a tile: its own frames (forward); EXT more on either side (backward: W_w z needs the neighbours)
  constexpr int NO = BWD ? TT + 2 * EXT : TT;
  constexpr int NB = NO + 2 * H;                   // right-hand-side rows under the taps
  constexpr int NF = BWD ? NB : NB + 2 * EXT;      // frames read (forward: b[i] needs mu of i - EXT .. i + EXT)
  constexpr int EW = BWD ? E + EXT : E;            // rows at either end that fir_ends writes
  constexpr int FB = 14;                           // frames per batch of loads
  constexpr int NQ = (NF + FB - 1) / FB;
  static_assert(NB % 2 == 0 && NO % 2 == 0, "row pairs");
  constexpr unsigned kDrop = 0x80000000u;          // an offset behind the buffer's end: the store is dropped
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long item = (long)blk * kW + wv;
  double *const part = a.partials + (size_t)blockIdx.x * kW + wv;  // (MSE, forward: every wavefront of the launch writes its own)
  if (item >= (long)a.nsg * a.nt) {
    if (MSE && !BWD && lane == 0) *part = 0.0;
    return;
  }
  const int g = (int)(item / a.nt), tile = (int)(item - (long)g * a.nt);
  const int b = g / a.ndg, dg = g - b * a.ndg;
  const int T = p.Tmax, sd = p.sd;
  const int d0 = dg * a.dgw;
  const int nd = sd - d0 < a.dgw ? sd - d0 : a.dgw;
  const bool lane_ok = lane < nd;
  const int d = d0 + (lane_ok ? lane : nd - 1);
  const int t0 = tile * TT;
  if (t0 + TT <= EW || t0 >= T - EW) {  // every row of the tile belongs to fir_ends
    if (MSE && !BWD && lane == 0) *part = 0.0;
    return;
  }
  const int nw = a.nw, mw = a.mw;
  const __amdgpu_buffer_rsrc_t irs = make_rsrc(BWD ? (const float *)p.grad_out + (size_t)b * T * p.ld_gout + d0
                                                   : (const float *)p.mean + (size_t)b * T * p.ld_in + d0);
  const __amdgpu_buffer_rsrc_t ors = make_rsrc((float *)p.out + (size_t)b * T * p.ld_out + d0);
  const unsigned loff = (unsigned)(d - d0) * 4u;
  const unsigned soff_ok = lane_ok ? loff : kDrop;
  const unsigned ldi_bytes = (unsigned)(BWD ? p.ld_gout : p.ld_in) * 4u, ldo_bytes = (unsigned)p.ld_out * 4u, win_bytes = (unsigned)sd * 4u;
  const int f_b = t0 - (BWD ? EXT : 0) - H;        // frame of right-hand-side row 0
  const int f_first = f_b - (BWD ? 0 : EXT);       // first frame read
  const float tapv = a.tap[lane < kTaps ? lane : 0];  // the interior row, tap k in lane k
  // window w applies to the frames lo[w] <= t < lo[w] + span[w] (one unsigned comparison per load / store)
  int lo[kMaxNw];
  unsigned span[kMaxNw];
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w) {
    lo[w] = w == 0 ? 0 : mw;
    span[w] = w >= nw ? 0u : (w == 0 ? (unsigned)T : (mw != 0 && T > 2 * mw ? (unsigned)(T - 2 * mw) : 0u));
  }
  float cw[kMaxNw][2 * EXT + 1];
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w)
#pragma unroll
    for (int k = 0; k <= 2 * EXT; ++k) cw[w][k] = a.cpad[w][k + kEXT - EXT];
  float bb[NB];  // the right-hand side under the taps: row ib <-> frame f_b + ib
  float out[NO];

  if (BWD) {
#pragma unroll
    for (int s = 0; s < NF; ++s) {
      const int t = f_first + s;
      const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
      bb[s] = ld_f32(irs, (unsigned)tc * ldi_bytes, (unsigned)t < (unsigned)T ? loff : kDrop);  // (a read behind the buffer's end returns 0)
    }
    __builtin_amdgcn_sched_barrier(0);
  } else {
    float mu[2][FB][kMaxNw];
#pragma unroll
    for (int i = 0; i < NB; ++i) bb[i] = 0.0f;
#pragma unroll
    for (int q = -1; q < NQ; ++q) {
      if (q + 1 < NQ) {  // the loads of batch q + 1
        // (opaque: else the scalar offsets and masks of all 252 loads are computed at the top of the kernel and spill)
        int fq = f_first + (q + 1) * FB;
        asm volatile("" : "+s"(fq));
#pragma unroll
        for (int s = 0; s < FB; ++s) {
          if ((q + 1) * FB + s >= NF) continue;
          const int t = fq + s;
          const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
          for (int w = 0; w < kMaxNw; ++w) {
            // the window's mask and the utterance's ends through the offset: a read behind the buffer's end returns 0
            const bool lv = (unsigned)(t - lo[w]) < span[w];
            mu[(q + 1) & 1][s][w] = ld_f32(irs, (unsigned)tc * ldi_bytes + (unsigned)(w < nw ? w : 0) * win_bytes, lv ? loff : kDrop);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q >= 0) {  // frame t = f_first + s spreads c_w[l + k] mu_w[t] over b[t + k]
#pragma unroll
        for (int sl = 0; sl < FB; ++sl) {
          const int s = q * FB + sl;
          if (s >= NF) continue;
#pragma unroll
          for (int w = 0; w < kMaxNw; ++w) {
            const float m = mu[q & 1][sl][w];
#pragma unroll
            for (int k = -EXT; k <= EXT; ++k) {
              const int ib = s - EXT + k;
              if (ib < 0 || ib >= NB) continue;
              if (w == 0 && k != 0) continue;  // (window 0 is a single tap: fir_shape_supported)
              bb[ib] = __builtin_fmaf(cw[w][k + EXT], m, bb[ib]);
            }
          }
          // b is only read by the taps below, and the compiler knows it: left alone it sinks every one of these sums down there
          // and keeps all the loaded values alive until then.  An opaque use pins the row this frame completes.
          if (s - 2 * EXT >= 0 && s - 2 * EXT < NB) asm volatile("" : "+v"(bb[s - 2 * EXT]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // The taps, two rows per instruction (v_pk_fma_f32): row pairs starting at an even row (be) and at an odd row (bo) are kept as
  // register pairs; tap k of the output pair (2 j, 2 j + 1) multiplies the pair starting at row 2 j + k.
  f32x2 be[NB / 2], bo[NB / 2 - 1], o2[NO / 2];
#pragma unroll
  for (int i = 0; i < NB / 2; ++i) {
    be[i] = f32x2{bb[2 * i], bb[2 * i + 1]};
    asm volatile("" : "+v"(be[i]));
  }
#pragma unroll
  for (int i = 0; i < NB / 2 - 1; ++i) {
    bo[i] = f32x2{bb[2 * i + 1], bb[2 * i + 2]};
    asm volatile("" : "+v"(bo[i]));
  }
#pragma unroll
  for (int j = 0; j < NO / 2; ++j) o2[j] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    const float tk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tapv), k));
    const f32x2 t2 = f32x2{tk, tk};
#pragma unroll
    for (int j = 0; j < NO / 2; ++j) o2[j] = __builtin_elementwise_fma(t2, (k & 1) ? bo[j + (k - 1) / 2] : be[j + k / 2], o2[j]);
  }
#pragma unroll
  for (int j = 0; j < NO / 2; ++j) {
    out[2 * j] = o2[j].x;
    out[2 * j + 1] = o2[j].y;
  }

  // (opaque, and behind the first output: else the offsets and masks of every store are computed at the top of the kernel and spill)
  int t0e = t0;
  asm volatile("" : "+s"(t0e), "+v"(out[0]));
  if (!BWD && MSE) {
    // y is stored if asked for; dy = scale (y - target) to the step's buffer; (y - target)^2 summed over this tile's own rows
    const __amdgpu_buffer_rsrc_t trs = make_rsrc(a.target + (size_t)b * T * sd + d0);
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(a.dy + (size_t)b * T * sd + d0);
    const unsigned yoff = a.y_given ? soff_ok : kDrop;
    float tg[TT];
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      tg[r] = ld_f32(trs, (unsigned)(t < T ? t : 0) * win_bytes, (unsigned)(t - EW) < (unsigned)(T - 2 * EW) ? soff_ok : kDrop);
    }
    __builtin_amdgcn_sched_barrier(0);
    int t0f = t0e;  // (opaque again: else the 32 row masks of the loads above are kept in scalar registers for the stores, and spill)
    asm volatile("" : "+s"(t0f));
    float ls = 0.0f;
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0f + r;
      const bool mine = (unsigned)(t - EW) < (unsigned)(T - 2 * EW);
      const unsigned so = mine ? soff_ok : kDrop;
      const float e = out[r] - tg[r];
      const float em = so == kDrop ? 0.0f : e;
      st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes, mine ? yoff : kDrop, out[r]);
      st_f32(drs, (unsigned)(t < T ? t : 0) * win_bytes, so, a.scale * e);
      ls = __builtin_fmaf(em, em, ls);
      if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const double tot = wave_sum((double)ls);
    if (lane == 0) *part = tot;
  } else if (!BWD) {
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes, (unsigned)(t - EW) < (unsigned)(T - 2 * EW) ? soff_ok : kDrop, out[r]);
      if (r % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // (else every offset is computed up front and the scalars spill)
    }
  } else {
    // grad[t, w] = m_w[t] sum_k c_w[l + k] z[t + k];  z of frame t0 + r + k is out[r + EXT + k]
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      const bool mine = (unsigned)(t - EW) < (unsigned)(T - 2 * EW);
#pragma unroll
      for (int w = 0; w < kMaxNw; ++w) {
        const bool lv = (unsigned)(t - lo[w]) < span[w];
        float gsum = 0.0f;
#pragma unroll
        for (int k = -EXT; k <= EXT; ++k) {
          if (w == 0 && k != 0) continue;
          gsum = __builtin_fmaf(cw[w][k + EXT], out[r + EXT + k], gsum);
        }
        st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes + (unsigned)(w < nw ? w : 0) * win_bytes, (mine && w < nw) ? soff_ok : kDrop, lv ? gsum : 0.0f);
      }
      if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // (else every offset is computed up front and the scalars spill)
    }
  }
}

// fir_tiles_shared (round 5: the forward launches' tile form, see shared_tiles(); parity 73 tests + 2 368 soak cases on the GPU,
// numpy model of its index logic: tools/experimental/fir_shared/model.py): the eight tiles of a workgroup are eight
// consecutive tiles of ONE (utterance, dim group) and share the right-hand side through LDS.  Phase A: wavefront v forms rows
// [v RPW, (v + 1) RPW) of the NROW = 8 * 32 + 2 H (+ 2 EXT backward) rows the group needs -- 40 frames x nw loads and 38 x 7 products
// instead of 84 x nw and 82 x 7 per tile; phase B: every wavefront reads its 80 (84) rows back as register PAIRS (rows are 256 bytes
// apart: ds_read2st64_b32 returns rows r, r + 1 in one instruction, so the even and the odd pairs cost no copies) and runs the taps.
template <bool BWD, int EXT, bool MSE>
__device__ __forceinline__ void fir_tiles_shared(const Problem &p, const Args &a, const unsigned blk) {
  constexpr int H = kH, E = kE, TT = kTT, NWV = kW;
  constexpr int NO = BWD ? TT + 2 * EXT : TT;
  constexpr int NB = NO + 2 * H;                   // right-hand-side rows under the taps of one tile
  constexpr int XB = BWD ? EXT : 0;
  constexpr int NROW = NWV * TT + 2 * H + 2 * XB;  // rows the group shares: frames F0 - H - XB .. F0 + 8 TT + H + XB - 1
  constexpr int RPW = (NROW + NWV - 1) / NWV;      // rows per wavefront in phase A
  constexpr int NFA = BWD ? RPW : RPW + 2 * EXT;   // frames a wavefront reads in phase A
  constexpr int EW = BWD ? E + EXT : E;
  constexpr int FB = 14;
  constexpr int NQ = (NFA + FB - 1) / FB;
  static_assert(NB % 2 == 0 && NO % 2 == 0, "row pairs");
  constexpr unsigned kDrop = 0x80000000u;
  __shared__ float lb[RPW * NWV][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ngt = (a.nt + NWV - 1) / NWV;          // groups of eight tiles per (utterance, dim group)
  const int g = (int)(blk / (unsigned)ngt), grp = (int)(blk - (unsigned)g * (unsigned)ngt);
  double *const part = a.partials + (size_t)blockIdx.x * kW + wv;
  const int b = g / a.ndg, dg = g - b * a.ndg;
  const int T = p.Tmax, sd = p.sd;
  const int d0 = dg * a.dgw;
  const int nd = sd - d0 < a.dgw ? sd - d0 : a.dgw;
  const bool lane_ok = lane < nd;
  const int d = d0 + (lane_ok ? lane : nd - 1);
  const int F0 = grp * NWV * TT;
  const int tile = grp * NWV + wv;
  const int t0 = tile * TT;
  const int nw = a.nw, mw = a.mw;
  const __amdgpu_buffer_rsrc_t irs = make_rsrc(BWD ? (const float *)p.grad_out + (size_t)b * T * p.ld_gout + d0
                                                   : (const float *)p.mean + (size_t)b * T * p.ld_in + d0);
  const __amdgpu_buffer_rsrc_t ors = make_rsrc((float *)p.out + (size_t)b * T * p.ld_out + d0);
  const unsigned loff = (unsigned)(d - d0) * 4u;
  const unsigned soff_ok = lane_ok ? loff : kDrop;
  const unsigned ldi_bytes = (unsigned)(BWD ? p.ld_gout : p.ld_in) * 4u, ldo_bytes = (unsigned)p.ld_out * 4u, win_bytes = (unsigned)sd * 4u;
  const float tapv = a.tap[lane < kTaps ? lane : 0];
  int lo[kMaxNw];
  unsigned span[kMaxNw];
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w) {
    lo[w] = w == 0 ? 0 : mw;
    span[w] = w >= nw ? 0u : (w == 0 ? (unsigned)T : (mw != 0 && T > 2 * mw ? (unsigned)(T - 2 * mw) : 0u));
  }
  float cw[kMaxNw][2 * EXT + 1];
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w)
#pragma unroll
    for (int k = 0; k <= 2 * EXT; ++k) cw[w][k] = a.cpad[w][k + kEXT - EXT];

  // ---- phase A: rows [wv RPW, (wv + 1) RPW) of the shared right-hand side; row r <-> frame F0 - H - XB + r ----
  {
    const int fA = F0 - H - XB + wv * RPW;         // frame of this wavefront's first row
    const int f_first = fA - (BWD ? 0 : EXT);      // first frame read
    if (BWD) {
      float v[RPW];
#pragma unroll
      for (int s = 0; s < RPW; ++s) {
        const int t = f_first + s;
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        v[s] = ld_f32(irs, (unsigned)tc * ldi_bytes, (unsigned)t < (unsigned)T ? loff : kDrop);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < RPW; ++s) lb[wv * RPW + s][lane] = v[s];
    } else {
      float bbA[RPW];
      float mu[2][FB][kMaxNw];
#pragma unroll
      for (int i = 0; i < RPW; ++i) bbA[i] = 0.0f;
#pragma unroll
      for (int q = -1; q < NQ; ++q) {
        if (q + 1 < NQ) {
          int fq = f_first + (q + 1) * FB;
          asm volatile("" : "+s"(fq));
#pragma unroll
          for (int s = 0; s < FB; ++s) {
            if ((q + 1) * FB + s >= NFA) continue;
            const int t = fq + s;
            const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
            for (int w = 0; w < kMaxNw; ++w) {
              const bool lv = (unsigned)(t - lo[w]) < span[w];
              mu[(q + 1) & 1][s][w] = ld_f32(irs, (unsigned)tc * ldi_bytes + (unsigned)(w < nw ? w : 0) * win_bytes, lv ? loff : kDrop);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (q >= 0) {
#pragma unroll
          for (int sl = 0; sl < FB; ++sl) {
            const int s = q * FB + sl;
            if (s >= NFA) continue;
#pragma unroll
            for (int w = 0; w < kMaxNw; ++w) {
              const float m = mu[q & 1][sl][w];
#pragma unroll
              for (int k = -EXT; k <= EXT; ++k) {
                const int ib = s - EXT + k;
                if (ib < 0 || ib >= RPW) continue;
                if (w == 0 && k != 0) continue;
                bbA[ib] = __builtin_fmaf(cw[w][k + EXT], m, bbA[ib]);
              }
            }
            // the row this frame completes goes to LDS (a store: nothing for the compiler to sink)
            if (s - 2 * EXT >= 0 && s - 2 * EXT < RPW) lb[wv * RPW + s - 2 * EXT][lane] = bbA[s - 2 * EXT];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __syncthreads();
  if (tile >= a.nt || t0 + TT <= EW || t0 >= T - EW) {  // no such tile, or every row of it belongs to fir_ends
    if (MSE && !BWD && lane == 0) *part = 0.0;
    return;
  }

  // ---- phase B: this tile's NB rows start at shared row wv TT; pairs from even and from odd rows ----
  float out[NO];
  f32x2 be[NB / 2], bo[NB / 2 - 1], o2[NO / 2];
#pragma unroll
  for (int i = 0; i < NB / 2; ++i) be[i] = f32x2{lb[wv * TT + 2 * i][lane], lb[wv * TT + 2 * i + 1][lane]};
#pragma unroll
  for (int i = 0; i < NB / 2 - 1; ++i) bo[i] = f32x2{lb[wv * TT + 2 * i + 1][lane], lb[wv * TT + 2 * i + 2][lane]};
#pragma unroll
  for (int j = 0; j < NO / 2; ++j) o2[j] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    const float tk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tapv), k));
    const f32x2 t2 = f32x2{tk, tk};
#pragma unroll
    for (int j = 0; j < NO / 2; ++j) o2[j] = __builtin_elementwise_fma(t2, (k & 1) ? bo[j + (k - 1) / 2] : be[j + k / 2], o2[j]);
  }
#pragma unroll
  for (int j = 0; j < NO / 2; ++j) {
    out[2 * j] = o2[j].x;
    out[2 * j + 1] = o2[j].y;
  }

  // (opaque, and behind the first output: else the offsets and masks of every store are computed at the top of the kernel and spill)
  int t0e = t0;
  asm volatile("" : "+s"(t0e), "+v"(out[0]));
  if (!BWD && MSE) {
    // y is stored if asked for; dy = scale (y - target) to the step's buffer; (y - target)^2 summed over this tile's own rows
    const __amdgpu_buffer_rsrc_t trs = make_rsrc(a.target + (size_t)b * T * sd + d0);
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(a.dy + (size_t)b * T * sd + d0);
    const unsigned yoff = a.y_given ? soff_ok : kDrop;
    float tg[TT];
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      tg[r] = ld_f32(trs, (unsigned)(t < T ? t : 0) * win_bytes, (unsigned)(t - EW) < (unsigned)(T - 2 * EW) ? soff_ok : kDrop);
    }
    __builtin_amdgcn_sched_barrier(0);
    int t0f = t0e;
    asm volatile("" : "+s"(t0f));
    float ls = 0.0f;
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0f + r;
      const bool mine = (unsigned)(t - EW) < (unsigned)(T - 2 * EW);
      const unsigned so = mine ? soff_ok : kDrop;
      const float e = out[r] - tg[r];
      const float em = so == kDrop ? 0.0f : e;
      st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes, mine ? yoff : kDrop, out[r]);
      st_f32(drs, (unsigned)(t < T ? t : 0) * win_bytes, so, a.scale * e);
      ls = __builtin_fmaf(em, em, ls);
      if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const double tot = wave_sum((double)ls);
    if (lane == 0) *part = tot;
  } else if (!BWD) {
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes, (unsigned)(t - EW) < (unsigned)(T - 2 * EW) ? soff_ok : kDrop, out[r]);
      if (r % 8 == 7) __builtin_amdgcn_sched_barrier(0);  // (else every offset is computed up front and the scalars spill)
    }
  } else {
    // grad[t, w] = m_w[t] sum_k c_w[l + k] z[t + k];  z of frame t0 + r + k is out[r + EXT + k]
#pragma unroll
    for (int r = 0; r < TT; ++r) {
      const int t = t0e + r;
      const bool mine = (unsigned)(t - EW) < (unsigned)(T - 2 * EW);
#pragma unroll
      for (int w = 0; w < kMaxNw; ++w) {
        const bool lv = (unsigned)(t - lo[w]) < span[w];
        float gsum = 0.0f;
#pragma unroll
        for (int k = -EXT; k <= EXT; ++k) {
          if (w == 0 && k != 0) continue;
          gsum = __builtin_fmaf(cw[w][k + EXT], out[r + EXT + k], gsum);
        }
        st_f32(ors, (unsigned)(t < T ? t : 0) * ldo_bytes + (unsigned)(w < nw ? w : 0) * win_bytes, (mine && w < nw) ? soff_ok : kDrop, lv ? gsum : 0.0f);
      }
      if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // (else every offset is computed up front and the scalars spill)
    }
  }
}

// fir_ends: one workgroup per (utterance, dim group, end): the EW rows at that end, whose taps come from the table.
// Rows are counted from the end (j = 0 is the first / last frame; the table's rows for the last frames are stored mirrored, so both
// ends run the same code).  Eight wavefronts share the rows: right-hand side and filter outputs through LDS, H rows of zeros in front
// of the right-hand side standing for the frames beyond the end.  No condition inside the loops.
template <bool BWD, int EXT, bool MSE>
__device__ __forceinline__ void fir_ends(const Problem &p, const Args &a, const unsigned blk) {
  constexpr int H = kH, E = kE, NWV = kW;
  constexpr int EW = BWD ? E + EXT : E;            // rows written
  constexpr int NZ = BWD ? EW + EXT : EW;          // FIR outputs needed (backward: z up to EXT beyond the last gradient row)
  constexpr int NBE = NZ + H;                      // right-hand-side rows needed (towards the interior; nothing beyond the end)
  __shared__ float lb[H + NBE][64];                // right-hand side: row H + j <-> the frame j from the end
  __shared__ float lz[NZ][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = (int)(blk >> 1);
  const bool bottom = blk & 1;
  const int b = g / a.ndg, dg = g - b * a.ndg;
  const int T = p.Tmax, sd = p.sd;
  const int d0 = dg * a.dgw;
  const int nd = sd - d0 < a.dgw ? sd - d0 : a.dgw;
  const bool lane_ok = lane < nd;
  const int d = d0 + (lane_ok ? lane : nd - 1);
  const int nw = a.nw, mw = a.mw;
  constexpr unsigned kDrop = 0x80000000u;          // an offset behind the buffer's end: the read returns 0
  const __amdgpu_buffer_rsrc_t irs = make_rsrc(BWD ? (const float *)p.grad_out + (size_t)b * T * p.ld_gout + d0
                                                   : (const float *)p.mean + (size_t)b * T * p.ld_in + d0);
  const unsigned loff = (unsigned)(d - d0) * 4u;
  const unsigned ldi_bytes = (unsigned)(BWD ? p.ld_gout : p.ld_in) * 4u, win_bytes = (unsigned)sd * 4u;
  float *obase = (float *)p.out + (size_t)b * T * p.ld_out + d;
  auto frame = [&](int j) { return bottom ? T - 1 - j : j; };
  float cb[kMaxNw][2 * EXT + 1];  // the windows, counted towards the interior
#pragma unroll
  for (int w = 0; w < kMaxNw; ++w)
#pragma unroll
    for (int k = 0; k <= 2 * EXT; ++k) cb[w][k] = bottom ? a.cpad[w][kEXT + EXT - k] : a.cpad[w][kEXT - EXT + k];
  // ---- the taps: row j uses the table row of its end for j < E, the interior row otherwise; tap k in lane k.  Requested first
  // (round 5: they depend on nothing; behind the barrier below they were a third dependent round trip -- measured, though, the
  // kernel is as fast either way, 12.1 vs 12.2 us at config 3: the table sits in L2) ----
  constexpr int NRZ = (NZ + NWV - 1) / NWV;
  float tv[NRZ];
#pragma unroll
  for (int i = 0; i < NRZ; ++i) {
    const int j = i * NWV + wv;
    const int jc = j < NZ ? j : NZ - 1;
    tv[i] = a.tap[(size_t)(jc < E ? (bottom ? 1 + E + jc : 1 + jc) : 0) * kTaps + (lane < kTaps ? lane : 0)];
  }
  // ---- right-hand side rows j = -H .. NBE - 1: every load first, then the sums ----
  constexpr int NR = (NBE + NWV - 1) / NWV;        // rows per wavefront
  constexpr int NL = BWD ? 1 : 1 + 2 * (2 * EXT + 1);
  float mu[NR][NL];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int j = i * NWV + wv;
    const int fi = frame(j < NBE ? j : NBE - 1);
    if (BWD) {
      mu[i][0] = ld_f32(irs, (unsigned)fi * ldi_bytes, loff);
    } else {
#pragma unroll
      for (int w = 0; w < kMaxNw; ++w) {
#pragma unroll
        for (int k = -EXT; k <= EXT; ++k) {   // b[i] += c_w[l + k] m_w[t] mu_w[t],  t = i - k
          if (w == 0 && k != 0) continue;
          const int t = fi - k;
          const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
          const bool lv = w < nw && t >= 0 && t < T && (w == 0 || (mw != 0 && t >= mw && t < T - mw));
          mu[i][w == 0 ? 0 : 1 + (w - 1) * (2 * EXT + 1) + k + EXT] =
              ld_f32(irs, (unsigned)tc * ldi_bytes + (unsigned)(w < nw ? w : 0) * win_bytes, lv ? loff : kDrop);
        }
      }
    }
  }
  for (int j = wv; j < H; j += NWV) lb[j][lane] = 0.0f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int j = i * NWV + wv;
    float v = mu[i][0];
    if (!BWD) {
      v *= a.cpad[0][kEXT];
#pragma unroll
      for (int w = 1; w < kMaxNw; ++w)
#pragma unroll
        for (int k = -EXT; k <= EXT; ++k) v = __builtin_fmaf(a.cpad[w][k + kEXT], mu[i][1 + (w - 1) * (2 * EXT + 1) + k + EXT], v);
    }
    if (j < NBE) lb[H + j][lane] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRZ; ++i) {
    const int j = i * NWV + wv;
    const int jc = j < NZ ? j : NZ - 1;
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {  // tap k multiplies the right-hand side of the frame j - H + k from the end
      const float tk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tv[i]), k));
      if (k & 1) s1 = __builtin_fmaf(tk, lb[jc + k][lane], s1);
      else s0 = __builtin_fmaf(tk, lb[jc + k][lane], s0);
    }
    if (j < NZ) lz[j][lane] = s0 + s1;
  }
  __syncthreads();
  if (MSE && !BWD) {
    float ls = 0.0f;
#pragma unroll
    for (int j0 = 0; j0 < EW; j0 += NWV) {
      const int j = j0 + wv;
      if (j >= EW) continue;
      const int t = frame(j);
      if (lane_ok) {
        const float y = lz[j][lane];
        const size_t at = ((size_t)b * T + t) * sd + d;
        const float e = y - a.target[at];
        if (a.y_given) obase[(size_t)t * p.ld_out] = y;
        a.dy[at] = a.scale * e;
        ls = __builtin_fmaf(e, e, ls);
      }
    }
    const double tot = wave_sum((double)ls);
    if (lane == 0) a.partials[(size_t)blockIdx.x * kW + wv] = tot;
    return;
  }
  if (!lane_ok) return;
#pragma unroll
  for (int j0 = 0; j0 < EW; j0 += NWV) {
    const int j = j0 + wv;
    if (j >= EW) continue;
    const int t = frame(j);
    if (!BWD) {
      obase[(size_t)t * p.ld_out] = lz[j][lane];
    } else {
#pragma unroll
      for (int w = 0; w < kMaxNw; ++w) {
        if (w >= nw) continue;
        const bool lv = w == 0 || (mw != 0 && t >= mw && t < T - mw);
        float gsum = 0.0f;
#pragma unroll
        for (int k = -EXT; k <= EXT; ++k) {  // z of the frame j + k from the end is z of frame t + k (first frames) or t - k (last)
          if (w == 0 && k != 0) continue;
          const int jj = j + k < 0 ? 0 : j + k;  // (beyond the end only where lv is false)
          gsum = __builtin_fmaf(cb[w][k + EXT], lz[jj][lane], gsum);
        }
        obase[(size_t)t * p.ld_out + (size_t)w * sd] = lv ? gsum : 0.0f;
      }
    }
  }
}

// One launch: the first 2 nsg workgroups take the ends (three dependent phases: they start first), the others eight tiles each.
// 64 utterances x 500 frames: 128 + 128 workgroups of eight wavefronts, one per CU.
// MSE (the training step, mlpg_hip_unit_mse_step): see Args; the backward launch has one more workgroup, which adds up the loss.
template <bool BWD, int EXT, bool MSE, bool SHARED = false>
__global__ __launch_bounds__(kW * 64, 1) void fir_kernel(const Problem p, const Args a) {
  const unsigned nends = 2u * (unsigned)a.nsg;
  if (MSE && BWD && blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < 64) {
      double s = 0.0;
      for (int i = (int)threadIdx.x; i < a.npart; i += 64) s += a.partials[i];
      s = wave_sum(s);
      if (threadIdx.x == 0) *a.loss = s * a.inv_n;
    }
    return;
  }
  if (blockIdx.x < nends) fir_ends<BWD, EXT, MSE>(p, a, blockIdx.x);
  else if (SHARED) fir_tiles_shared<BWD, EXT, MSE>(p, a, blockIdx.x - nends);
  else fir_tiles<BWD, EXT, MSE>(p, a, blockIdx.x - nends);
}

// ---- the tap table: once per (device, window set) ----
struct Table {
  float *dev = nullptr;
  bool ok = false;
};
std::mutex g_mu;
std::map<std::pair<int, std::vector<double>>, Table> g_tables;

std::vector<double> key_of(const WinSet &ws) {
  std::vector<double> k;
  k.push_back(ws.nw);
  for (int w = 0; w < ws.nw; ++w) {
    k.push_back(ws.l[w]);
    k.push_back(ws.u[w]);
    for (int j = 0; j <= ws.l[w] + ws.u[w]; ++j) k.push_back(ws.c[ws.off[w] + j]);
  }
  return k;
}

// Builds the table with one exact forward solve of 1 + 2 E one-hot static-window inputs on a kTref-frame utterance (b = c_0 e_s, so
// the trajectory is c_0 times column s of P^-1).  Synchronous (first use only); not while `st` is being captured.
const Table *table_for(hipStream_t st, int device, const WinSet &ws) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair(device, key_of(ws));
  auto it = g_tables.find(key);
  if (it != g_tables.end()) return &it->second;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // (not cached: a later call outside the capture builds it)
  }
  Table tb;
  const int S = kRows, T = kTref, D = ws.nw * S;
  const double c0 = ws.c[ws.off[0]];
  std::vector<double> means((size_t)T * D, 0.0), outh((size_t)T * S, 0.0);
  auto src_frame = [&](int dcol) { return dcol == 0 ? T / 2 : (dcol <= kE ? dcol - 1 : T - 1 - (dcol - kE - 1)); };
  for (int dcol = 0; dcol < S; ++dcol) means[(size_t)src_frame(dcol) * D + dcol] = 1.0 / c0;
  double *dm = nullptr, *dout = nullptr;
  bool good = hipMalloc(&dm, means.size() * 8) == hipSuccess && hipMalloc(&dout, outh.size() * 8) == hipSuccess;
  if (good) good = hipMemcpyAsync(dm, means.data(), means.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess;
  if (good) {
    Problem q;
    q.mean = dm;
    q.var = nullptr;
    q.grad_out = nullptr;
    q.lengths = nullptr;
    q.out = dout;
    q.status = nullptr;
    q.var_mode = MLPG_HIP_VAR_UNIT;
    q.B = 1;
    q.Tmax = T;
    q.D = D;
    q.sd = S;
    q.ld_in = D;
    q.ld_gout = 0;
    q.ld_out = S;
    q.ld_status = S;
    good = dispatch_solve(st, MLPG_HIP_F64, MLPG_HIP_F64, MLPG_HIP_ALGO_GENERIC, false, q, ws, device) == 0;
  }
  if (good) good = hipMemcpyAsync(outh.data(), dout, outh.size() * 8, hipMemcpyDeviceToHost, st) == hipSuccess;
  if (good) good = hipStreamSynchronize(st) == hipSuccess;
  if (good) {
    // column dcol of outh = column src_frame(dcol) of P^-1 = (symmetry) that ROW
    auto pinv = [&](int dcol, int s) -> double { return (s >= 0 && s < T) ? outh[(size_t)s * S + dcol] : 0.0; };
    std::vector<float> tap((size_t)kRows * kTaps, 0.0f);
    const int c = T / 2;
    for (int k = 0; k < kTaps; ++k) tap[k] = (float)pinv(0, c - kH + k);
    for (int t = 0; t < kE; ++t)
      for (int k = 0; k < kTaps; ++k) {
        tap[(size_t)(1 + t) * kTaps + k] = (float)pinv(1 + t, t - kH + k);
        tap[(size_t)(1 + kE + t) * kTaps + k] = (float)pinv(1 + kE + t, (T - 1 - t) + kH - k);  // mirrored: counted from the end
      }
    // the decay the kernel relies on, checked on this window set's own numbers
    const double g0 = std::abs(pinv(0, c)), tol = 0x1p-26 * g0;
    double tail = 0.0;
    for (int s = kH + 1; s < 2 * kH; ++s) tail += std::abs(pinv(0, c + s)) + std::abs(pinv(0, c - s));
    bool ok = g0 > 0.0 && std::abs(pinv(0, c + kH + 1)) <= tol && std::abs(pinv(0, c - kH - 1)) <= tol && tail <= 4.0 * tol;
    // the last rows of the end tables have reached the interior row
    for (int k = 0; k < kTaps && ok; ++k) {
      ok = std::abs((double)tap[(size_t)kE * kTaps + k] - (double)tap[k]) <= 64.0 * tol &&
           std::abs((double)tap[(size_t)(2 * kE) * kTaps + k] - (double)tap[k]) <= 64.0 * tol;
    }
    for (float v : tap) ok = ok && std::isfinite(v);
    if (ok) {
      ok = hipMalloc(&tb.dev, tap.size() * sizeof(float)) == hipSuccess &&
           hipMemcpy(tb.dev, tap.data(), tap.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    tb.ok = ok;
  }
  (void)hipGetLastError();
  if (dm) (void)hipFree(dm);
  if (dout) (void)hipFree(dout);
  if (!good) return nullptr;  // a runtime failure is not a property of the window set: not cached
  return &(g_tables[key] = tb);
}

}  // namespace fir

// float32 in and out, unit variances, no ragged lengths, at least 2 E + 2 H frames, 1-3 windows of extent <= 2 the first of which is a
// single tap (the static window), dense dims.
bool fir_shape_supported(const Problem &p, const WinSet &ws, int in_dtype, int out_dtype) {
  using namespace fir;
  if (in_dtype != MLPG_HIP_F32 || out_dtype != MLPG_HIP_F32 || p.var_mode != MLPG_HIP_VAR_UNIT || p.lengths) return false;
  if (ws.nw < 1 || ws.nw > kMaxNw || ws.mw > kEXT || ws.l[0] != 0 || ws.u[0] != 0 || ws.c[ws.off[0]] == 0.0) return false;
  if (p.pitch && p.pitch != p.sd) return false;
  if (p.Tmax < 2 * kE + 2 * kH || p.B < 1 || p.sd < 1) return false;
  return rows_fit_buffer(p);
}

// Does the FIR form serve this window set on this device: the tap table exists (it is built here, synchronously, on first use -- not
// while `st` is being captured) and passed its decay test.  The ONE answer both mlpg_hip_unit_mse_step and mlpg_hip_unit_mse_form give.
bool fir_table_ready(hipStream_t st, int device, const WinSet &ws) {
  const fir::Table *tb = fir::table_for(st, device, ws);
  return tb && tb->ok;
}

// Measured (profiles/r04_notes.md section 11): backward it is the fastest kernel at 64 and at 256 utterances; forward the
// constant-coefficient kernel (one workgroup per sequence) overtakes it once there are sequences enough to fill the chip with those.
// mlpg_hip_shutdown: the tap tables
void fir_shutdown() {
  std::lock_guard<std::mutex> lk(fir::g_mu);
  for (auto &kv : fir::g_tables) {
    if (!kv.second.dev) continue;
    (void)hipSetDevice(kv.first.first);
    (void)hipFree(kv.second.dev);
  }
  fir::g_tables.clear();
}

bool fir_preferred(const Problem &p, bool backward) {
  return backward || (long)p.B * ((p.sd + 63) / 64) < 256;
}

namespace fir {
void fill_args(Args &a, const float *tap, const Problem &p, const WinSet &ws) {
  a.tap = tap;
  a.ndg = (p.sd + 63) / 64;
  a.dgw = (p.sd + a.ndg - 1) / a.ndg;
  a.nsg = p.B * a.ndg;
  a.nt = (p.Tmax + kTT - 1) / kTT;
  a.nw = ws.nw;
  a.mw = ws.mw;
  for (int w = 0; w < kMaxNw; ++w) {
    for (int j = 0; j <= 2 * kEXT; ++j) a.cpad[w][j] = 0.0f;
    if (w >= ws.nw) continue;
    for (int k = -ws.l[w]; k <= ws.u[w]; ++k) a.cpad[w][k + kEXT] = (float)ws.c[ws.off[w] + ws.l[w] + k];
  }
  a.target = nullptr;
  a.dy = nullptr;
  a.partials = nullptr;
  a.loss = nullptr;
  a.scale = 0.0f;
  a.inv_n = 0.0;
  a.npart = 0;
  a.y_given = 1;
}
// Which tile form a launch takes.  Measured in round 5 (profiles/r05_notes.md; config 3 / 256 x 1000 x 60, kernel durations):
// the tiles of a workgroup sharing the right-hand side through LDS (fir_tiles_shared) take the forward pass from 17.0 to 13.5 us
// / 0.083 to 0.072 ms, the backward pass gains nothing (13.9 -> 13.7 us) or loses (0.062 -> 0.066 ms): its right-hand side is one
// row per frame, there is little to share.  Default: forward shared, backward not.  MLPG_FIR_SHARED=0 / 1 forces one form for
// both (A/B runs).
bool shared_tiles(bool backward) {
  static const int mode = [] {
    const char *e = std::getenv("MLPG_FIR_SHARED");
    return !e ? -1 : (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1));
  }();
  return mode < 0 ? !backward : mode == 1;
}
unsigned grid_of(const Args &a, bool shared) {
  if (shared) return (unsigned)(2 * a.nsg + (long)a.nsg * ((a.nt + kW - 1) / kW));
  return (unsigned)(2 * a.nsg + ((long)a.nsg * a.nt + kW - 1) / kW);
}
size_t align128(size_t n) { return (n + 127) / 128 * 128; }
}  // namespace fir

// The training step's workspace in this form: [128 bytes: the wave-per-system kernel's counter, untouched][one double per wavefront of
// the forward launch][dy, (B, Tmax, sd) float32].
size_t fir_mse_workspace_bytes(int B, int Tmax, int sd) {
  using namespace fir;
  const long ndg = (sd + 63) / 64, nsg = B * ndg, nt = (Tmax + kTT - 1) / kTT;
  const size_t npart = (size_t)(2 * nsg + nsg * ((nt + kW - 1) / kW)) * kW;  // (the larger of the two tile mappings)
  return 128 + align128(npart * sizeof(double)) + align128((size_t)B * Tmax * sd * sizeof(float));
}

// forward + MSE + backward of mlpg_hip_unit_mse_step as two launches (p: mean, ld_in = D; out = grad_mean, ld_out = D).
int launch_fir_mse(hipStream_t st, const Problem &p, const WinSet &ws, int device, const void *target, void *y_out, double n_elems,
                   double *loss, void *workspace) {
  using namespace fir;
  const Table *tb = table_for(st, device, ws);
  if (!tb || !tb->ok) return kFirNotApplicable;
  Args a;
  fill_args(a, tb->dev, p, ws);
  const bool shf = shared_tiles(false), shb = shared_tiles(true);
  const unsigned nblk = grid_of(a, shf), nblk_b = grid_of(a, shb);
  a.npart = (int)(nblk * kW);  // one partial sum per wavefront of the FORWARD launch
  a.target = (const float *)target;
  a.partials = (double *)((char *)workspace + 128);
  a.dy = (float *)((char *)workspace + 128 + align128((size_t)a.npart * sizeof(double)));
  a.loss = loss;
  a.scale = (float)(2.0 / n_elems);
  a.inv_n = 1.0 / n_elems;
  a.y_given = y_out ? 1 : 0;
  Problem pf = p, pb = p;
  pf.out = y_out;
  pf.ld_out = p.sd;
  pb.mean = nullptr;
  pb.grad_out = a.dy;
  pb.ld_gout = p.sd;
  note_launch(kCountFir);
  note_launch(kCountFir);
  const dim3 block(kW * 64);
  // (the backward launch's last workgroup adds up the forward launch's partial sums)
  if (ws.mw <= 1) {
    if (shf) hipLaunchKernelGGL((fir_kernel<false, 1, true, true>), dim3(nblk), block, 0, st, pf, a);
    else hipLaunchKernelGGL((fir_kernel<false, 1, true, false>), dim3(nblk), block, 0, st, pf, a);
    if (shb) hipLaunchKernelGGL((fir_kernel<true, 1, true, true>), dim3(nblk_b + 1), block, 0, st, pb, a);
    else hipLaunchKernelGGL((fir_kernel<true, 1, true, false>), dim3(nblk_b + 1), block, 0, st, pb, a);
  } else {
    if (shf) hipLaunchKernelGGL((fir_kernel<false, 2, true, true>), dim3(nblk), block, 0, st, pf, a);
    else hipLaunchKernelGGL((fir_kernel<false, 2, true, false>), dim3(nblk), block, 0, st, pf, a);
    if (shb) hipLaunchKernelGGL((fir_kernel<true, 2, true, true>), dim3(nblk_b + 1), block, 0, st, pb, a);
    else hipLaunchKernelGGL((fir_kernel<true, 2, true, false>), dim3(nblk_b + 1), block, 0, st, pb, a);
  }
  MLPG_HIP_CHECK(hipGetLastError());
  if (p.status) MLPG_HIP_CHECK(hipMemset2DAsync(p.status, (size_t)p.ld_status * sizeof(int32_t), 0, (size_t)p.sd * sizeof(int32_t), (size_t)p.B, st));
  return 0;
}

int launch_fir(hipStream_t st, bool backward, const Problem &p, const WinSet &ws, int device) {
  using namespace fir;
  const Table *tb = table_for(st, device, ws);
  if (!tb || !tb->ok) return kFirNotApplicable;
  Args a;
  fill_args(a, tb->dev, p, ws);
  note_launch(kCountFir);
  const bool shared = shared_tiles(backward);
  const dim3 grid(grid_of(a, shared)), block(kW * 64);
  if (shared) {
    if (ws.mw <= 1) {
      if (backward) hipLaunchKernelGGL((fir_kernel<true, 1, false, true>), grid, block, 0, st, p, a);
      else hipLaunchKernelGGL((fir_kernel<false, 1, false, true>), grid, block, 0, st, p, a);
    } else {
      if (backward) hipLaunchKernelGGL((fir_kernel<true, 2, false, true>), grid, block, 0, st, p, a);
      else hipLaunchKernelGGL((fir_kernel<false, 2, false, true>), grid, block, 0, st, p, a);
    }
  } else if (ws.mw <= 1) {  // (extent 0 or 1: the instance with 3 instead of 5 taps per window)
    if (backward) hipLaunchKernelGGL((fir_kernel<true, 1, false>), grid, block, 0, st, p, a);
    else hipLaunchKernelGGL((fir_kernel<false, 1, false>), grid, block, 0, st, p, a);
  } else {
    if (backward) hipLaunchKernelGGL((fir_kernel<true, 2, false>), grid, block, 0, st, p, a);
    else hipLaunchKernelGGL((fir_kernel<false, 2, false>), grid, block, 0, st, p, a);
  }
  MLPG_HIP_CHECK(hipGetLastError());
  // (P = c_0^2 I + a sum of squares is positive definite whatever the windows: every verdict is 0)
  if (p.status) MLPG_HIP_CHECK(hipMemset2DAsync(p.status, (size_t)p.ld_status * sizeof(int32_t), 0, (size_t)p.sd * sizeof(int32_t), (size_t)p.B, st));
  return 0;
}

}  // namespace mlpg
