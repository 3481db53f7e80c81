// Constant-coefficient MLPG kernels (algo = MLPG_HIP_ALGO_CONST; the AUTO choice for global (D,) and unit variances
// when the launch has enough (utterance, dim group) sequences to fill the chip).
//
// With global or unit variances the precision matrix  P_d = sum_w W_w^T diag(tau_w) W_w  of a static dim depends on
// (d, T) only -- the reference tiles the (D,) variances over the frames (paramgen/_mlpg.py:169-170) and re-factorises
// the same matrix for every utterance; unit_variance_mlpg_matrix (:297-373) and autograd.UnitVarianceMLPG
// (autograd/_impl/mlpg.py:70-172) are the same matrix for every dim as well.  Here it is factorised ONCE per launch:
//
//   setup_kernel   one wavefront per dim group, lane = static dim: natural-order LDL^T of P_d for T = infinity, row by
//                  row, until the multipliers have converged to their steady state (row i_s, a few dozen frames for
//                  ordinary variances); table[i] = (l1_i, l2_i, 1/d_i, d_i), i <= i_s.
//   stream_kernel  one workgroup walks one (utterance, dim group) sequence from its first frame to its last; lane =
//                  static dim (every load/store moves one contiguous run of sd elements of a frame), wavefront = chunk of
//                  M frames, the W chunks of a super-step (W M frames) side by side.  Forward and backward substitution
//                      z_i = b_i - l1_i z_{i-1} - l2_i z_{i-2},      y_i = z_i / d_i - l1_{i+1} y_{i+1} - l2_{i+2} y_{i+2}
//                  are second-order linear recurrences with data-independent coefficients: per-lane CONSTANTS in
//                  registers for every chunk inside the steady range (no reciprocal, no table), table rows for the
//                  first i_s frames, and the last two rows of an utterance (the reference zeroes the dynamic precisions
//                  on the last frame, :191-193) re-derived on the fly from the table's state.  A chunk runs the
//                  recurrences with zero incoming state and hands over (g, A): its end state for zero input and the
//                  2x2 transfer matrix of its rows, through LDS.  The forward state entering a super-step is carried
//                  from the one before (exact).  The backward recurrence needs the future: super-step k is finished
//                  ONE super-step late, from the upward state of super-step k+1 solved with zero input from below --
//                  what that leaves out is  B_{k+1} t_{k+2},  the transfer matrix of a whole super-step (its largest
//                  entry is ~1e-25 .. 1e-50 for ordinary variances), and the setup kernel checks it against kTol = 2^-56 for
//                  every dim of the group.  Where the check fails (dynamic features hundreds of times tighter than the
//                  static ones: decay over hundreds of frames) the sequence takes the exact two-sweep path: forward
//                  sweep with the scaled z parked in the output array, backward sweep in reverse order.
//                  No inter-workgroup traffic of any kind, no atomics, no flags; loads of the next super-step stream
//                  under the arithmetic of this one.
//                  Systems whose factor met a non-positive pivot get the reference's verdict (natural-order first failing
//                  pivot, zero column) from the workgroup that walked their sequence.
// tools/const_model.py is the executable specification of the chunk algebra (tests/test_const_model.py pins it against
// the oracle).  Windows: extents <= 1 with at least one dynamic window of extent 1 (mw == 1), NW = 2 or 3.
#pragma once
#include <string.h>

#include <mutex>
#include <type_traits>
#include <vector>
#include "assemble.h"

#ifdef MLPG_CONST_TIMING
#define CST_TICK(k)                                                    \
  do {                                                                 \
    const long long t_now_ = (long long)__builtin_readcyclecounter(); \
    tq[k] += t_now_ - t_prev;                                          \
    t_prev = t_now_;                                                   \
  } while (0)
#else
#define CST_TICK(k) do {} while (0)
#endif

namespace mlpg {
namespace cst {

// Largest entry allowed in a transfer matrix whose product with a state (a pair of trajectory values) is left out: the
// term then is below a quarter of an ulp of the largest trajectory value -- it cannot be told from rounding.
constexpr double kTol = 0x1p-56;
constexpr int kTabHead = 16;        // doubles per lane ahead of the table rows (see setup_kernel)

struct Args {
  double *tab;         // [dim group][kTabHead + 4 * tab_rows][64]
  int *tabi;           // [dim group][128]: [0] i_s, [1] lag ok, [64 + lane] first failing row + 1 of the T = infinity factor (0: none)
  double *key;         // [dim group][kMaxWindows + 2][64]: what the group's table was computed from (setup_kernel)
  int ndg, dgw, nsg, tab_rows;
  int stagger;         // start offset of every other group of 8 workgroups, in units of 8128 cycles (0: none); set by the launcher
                       // from the grid and the device's CU count (stream_kernel)
  // per window cm = W[t,t-1], c0 = W[t,t], cp = W[t,t+1] and the products cp cp, c0 c0, cm cm, cp c0, c0 cm, cp cm: computed
  // on the host so that the kernel holds them in scalar registers (a product of two scalar doubles formed in the kernel
  // is a vector instruction whose result stays in vector registers for the whole sequence loop)
  double wc[kMaxWindows][9];
  StreamMap sm;        // MULTI kernels only: the streams whose static dims sit side by side on the lanes
};

// MULTI kernels: the stream a merged static-dim index belongs to (at most 4 streams, begin[] ascending, unused entries =
// INT_MAX) and the dim's columns there (as strip::lane_stream)
struct LaneStream { int sd, din, dstat, dout; };
__device__ __forceinline__ LaneStream lane_stream(const StreamMap &sm, int d) {
  const int s_ = (d >= sm.begin[1]) + (d >= sm.begin[2]) + (d >= sm.begin[3]);
  auto pick = [&](const int (&v)[4]) { return s_ == 0 ? v[0] : s_ == 1 ? v[1] : s_ == 2 ? v[2] : v[3]; };
  const int dl = d - pick(sm.begin);
  return {pick(sm.sd), dl + pick(sm.in_col), dl + pick(sm.stat_col), dl + pick(sm.out_col)};
}

struct V2 { double x, y; };
struct M2 { double a, b, c, d; };  // [a b; c d]
__device__ __forceinline__ V2 mv(const M2 &A, const V2 &v) { return {A.a * v.x + A.b * v.y, A.c * v.x + A.d * v.y}; }
__device__ __forceinline__ V2 add(const V2 &a, const V2 &b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ M2 mm(const M2 &A, const M2 &B) {
  return {A.a * B.a + A.b * B.c, A.a * B.b + A.b * B.d, A.c * B.a + A.d * B.c, A.c * B.b + A.d * B.d};
}
__device__ __forceinline__ double amax4(const M2 &m) {
  return __builtin_fmax(__builtin_fmax(__builtin_fabs(m.a), __builtin_fabs(m.b)),
                        __builtin_fmax(__builtin_fabs(m.c), __builtin_fabs(m.d)));
}

__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}
template <typename T>
__device__ __forceinline__ double tau_of(T v) { return recip_in_dtype<T>(v); }  // 1/var in the input dtype (_mlpg.py:188)

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// ---- buffer loads / stores: wave-uniform descriptor, row offset in an SGPR, this lane's byte offset in one VGPR ----
template <typename T>
__device__ __forceinline__ T ld_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
// the same past the CU's L1 (sc0 sc1): rows another wavefront of the workgroup has just written
template <typename T>
__device__ __forceinline__ T ld_row_fresh(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
template <>
__device__ __forceinline__ double ld_row_fresh<double>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 17);
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <>
__device__ __forceinline__ float ld_row_fresh<float>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 17));
}
template <>
__device__ __forceinline__ double ld_row<double>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 0);
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <>
__device__ __forceinline__ float ld_row<float>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 0));
}
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const u32x2 w = {(unsigned)u, (unsigned)(u >> 32)};
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, loff, soff, 0);
}
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, loff, soff, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
  // the base must be wave-uniform PROVABLY (a lane-tainted descriptor is wrapped in a waterfall loop per access)
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// ---- the matrix: rows of P for one static dim -------------------------------------------------------------------
// live(w, t): frame t carries precision for window w -- [0, T) for the static window, [1, T-1) for the dynamic
// ones (mw == 1: _mlpg.py:177,191-193).  Wave-uniform.
__device__ __forceinline__ double live(int w, int t, int T) {
  return (w == 0 ? (t >= 0 && t < T) : (t >= 1 && t < T - 1)) ? 1.0 : 0.0;
}
// a = P[i,i], c = P[i,i-1], e = P[i,i-2] (entries outside the matrix are zero)
template <int NW>
__device__ __forceinline__ void p_entries(int i, int T, const double (&tau)[NW], const double (*wc)[9], double &a, double &c,
                                          double &e) {
  a = c = e = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const double tm = live(w, i - 1, T) * tau[w], t0 = live(w, i, T) * tau[w], tp = live(w, i + 1, T) * tau[w];
    a += tm * wc[w][3] + t0 * wc[w][4] + tp * wc[w][5];
    c += tm * wc[w][6] + t0 * wc[w][7];
    e += tm * wc[w][8];
  }
  if (i < 1) c = 0.0;
  if (i < 2) e = 0.0;
}
// one row of the pentadiagonal LDL^T from the state of the two rows above
struct FacState { double d1, dinv1, d2, dinv2, l1p; };  // d_{i-1}, 1/d_{i-1}, d_{i-2}, 1/d_{i-2}, l1_{i-1}
struct FacRow { double l1, l2, dinv, d; };
__device__ __forceinline__ FacRow ldl_row(double a, double c, double e, const FacState &s) {
  FacRow r;
  r.l2 = e * s.dinv2;
  r.l1 = (c - e * s.l1p) * s.dinv1;
  r.d = a - (r.l1 * r.l1) * s.d1 - (r.l2 * r.l2) * s.d2;
  r.dinv = fast_rcp(r.d);
  return r;
}
__device__ __forceinline__ FacState advance(const FacState &s, const FacRow &r) { return {r.d, r.dinv, s.d1, s.dinv1, r.l1}; }

// (d: the dim's window-0 column, sd: its window pitch -- the stream's own for a merged launch)
template <typename TIN, int VM, int NW>
__device__ __forceinline__ void lane_taus(const Problem &p, int d, int sd, double (&tau)[NW]) {
#pragma unroll
  for (int w = 0; w < NW; ++w) tau[w] = VM == MLPG_HIP_VAR_GLOBAL ? tau_of<TIN>(((const TIN *)p.var)[w * sd + d]) : 1.0;
}

// ---- setup: the T = infinity factor of every dim, until steady ---------------------------------------------------
// Table of one dim group, per lane: head[kTabHead] = A_inf (4), B_inf (4): transfer matrices of a chunk of M steady
// rows; l1, l2, 1/d of the steady state; tau_w (3); then rows[i] = (l1_i, l2_i, 1/d_i, d_i), i <= i_s.
// tabi[1] = 1 if the transfer matrix of a steady super-step (W chunks) is below kTol for every dim of the group and the
// transient ends inside the first super-step: stream_kernel then takes the one-step lag (each chunk decides from its own
// matrices whether it has to wait at all).
enum { hA = 0, hB = 4, hL1 = 8, hL2 = 9, hDinv = 10, hTau = 11 };
template <typename TIN, int VM, int NW, bool MULTI = false>
__global__ __launch_bounds__(64) void setup_kernel(Problem p, Args a, int M, int W, int fresh) {
  const int lane = threadIdx.x, dg = blockIdx.x;
  const int d0 = dg * a.dgw;
  const int sd_all = MULTI ? a.sm.total : p.sd;
  const int nd = sd_all - d0 < a.dgw ? sd_all - d0 : a.dgw;
  const int d = d0 + (lane < nd ? lane : nd - 1);
  double tau[NW];
  if (MULTI) {
    const LaneStream ls = lane_stream(a.sm, d);
    lane_taus<TIN, VM, NW>(p, ls.din, ls.sd, tau);
  } else {
    lane_taus<TIN, VM, NW>(p, d, p.sd, tau);
  }
  // The table is a function of the precisions, the window coefficients and the shape only: a launch that finds them
  // unchanged (the same global variances applied to batch after batch) leaves the table as it is.  `fresh`: the host
  // knows the scratch holds nothing yet.
  {
    double *key = a.key + (size_t)dg * (kMaxWindows + 2) * 64 + lane;
    bool same = !fresh;
#pragma unroll
    for (int w = 0; w < NW; ++w) same &= __double_as_longlong(key[w * 64]) == __double_as_longlong(tau[w]);
    // (everything the scratch layout and the table depend on, exact in 53 bits)
    const double shape = (double)((((((long long)a.tab_rows * 1024 + a.ndg) * 128 + a.dgw) * 64 + M) * 32 + W) * 8 + NW);
    same &= key[kMaxWindows * 64] == shape;
    double csum = 0.0;  // the window coefficients, folded per lane (lane l: coefficient l)
    if (lane < NW * 3) csum = a.wc[lane / 3][lane % 3];
    same &= __double_as_longlong(key[(kMaxWindows + 1) * 64]) == __double_as_longlong(csum);
    if (__ballot(!same) == 0ull) return;
#pragma unroll
    for (int w = 0; w < NW; ++w) key[w * 64] = tau[w];
    key[kMaxWindows * 64] = shape;
    key[(kMaxWindows + 1) * 64] = csum;
  }
  double *tab = a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + lane;
  double *rows = tab + kTabHead * 64;
  FacState s = {1.0, 1.0, 1.0, 1.0, 0.0};
  FacRow prev = {0.0, 0.0, 1.0, 1.0};
  int kfail = 0, run = 0, i = 0;
  const int big = 0x3fffffff;
  double ia, ic, ie;  // the interior row of P (rows >= 2 of the T = infinity matrix)
  p_entries<NW>(8, big, tau, a.wc, ia, ic, ie);
  for (; i < a.tab_rows; ++i) {
    double pa = ia, pc = ic, pe = ie;
    if (i < 2) p_entries<NW>(i, big, tau, a.wc, pa, pc, pe);
    const FacRow r = ldl_row(pa, pc, pe, s);
    if (!(r.d > 0.0) && kfail == 0) kfail = i + 1;
    rows[(4 * i + 0) * 64] = r.l1;
    rows[(4 * i + 1) * 64] = r.l2;
    rows[(4 * i + 2) * 64] = r.dinv;
    rows[(4 * i + 3) * 64] = r.d;
    // steady: the row repeats the previous one to 2^-50 (the recurrence contracts: it then stays put); failing lanes
    // do not hold the others up
    const double eps = 0x1p-50;
    const bool same = i >= 3 && __builtin_fabs(r.d - prev.d) <= eps * __builtin_fabs(r.d) &&
                      __builtin_fabs(r.l1 - prev.l1) <= eps * __builtin_fabs(r.l1) + 1e-300 &&
                      __builtin_fabs(r.l2 - prev.l2) <= eps * __builtin_fabs(r.l2) + 1e-300;
    run = __ballot(!(same || kfail != 0)) == 0ull ? run + 1 : 0;
    s = advance(s, r);
    prev = r;
    if (run >= 2) break;
  }
  const int i_s = i < a.tab_rows ? i : a.tab_rows - 1;
  // transfer matrices of a chunk of M steady rows: forward (columns = responses to s = (1,0), (0,1)) and backward
  double h1a = 1.0, h1b = 0.0, h2a = 0.0, h2b = 1.0;
  for (int k = 0; k < M; ++k) {
    const double n1 = -prev.l1 * h1a - prev.l2 * h1b, n2 = -prev.l1 * h2a - prev.l2 * h2b;
    h1b = h1a; h1a = n1;
    h2b = h2a; h2a = n2;
  }
  // the two recurrences have the same coefficients in the steady range: B_inf = A_inf
  tab[(hA + 0) * 64] = h1a; tab[(hA + 1) * 64] = h2a; tab[(hA + 2) * 64] = h1b; tab[(hA + 3) * 64] = h2b;
  tab[(hB + 0) * 64] = h1a; tab[(hB + 1) * 64] = h2a; tab[(hB + 2) * 64] = h1b; tab[(hB + 3) * 64] = h2b;
  tab[hL1 * 64] = prev.l1; tab[hL2 * 64] = prev.l2; tab[hDinv * 64] = prev.dinv;
#pragma unroll
  for (int w = 0; w < NW; ++w) tab[(hTau + w) * 64] = tau[w];
  // the transfer matrix of a steady super-step (what the one-step lag of stream_kernel leaves out, times the state
  // below that super-step): the chunk's to the power W
  M2 P = {h1a, h2a, h1b, h2b};
  M2 Q = {1.0, 0.0, 0.0, 1.0};
  for (int q = 0; q < W; ++q) Q = mm(P, Q);
  const bool small = amax4(Q) < kTol;  // (NaN: false)
  const int ok = __ballot(lane < nd && !small && kfail == 0) == 0ull && i < a.tab_rows && i_s + 2 < M * W;
  if (lane == 0) {
    a.tabi[dg * 128] = i_s;
    a.tabi[dg * 128 + 1] = ok;
  }
  a.tabi[dg * 128 + 64 + lane] = kfail;
}

// ---- per-row coefficients ----------------------------------------------------------------------------------------
// Chunks are aligned to the utterance's END: chunk j of NC = ceil(T / M) covers rows T - (NC - j) M .. + M - 1, so that
// the two rows whose factor differs from the table's (T-2, T-1) always are rows M-2, M-1 of the LAST chunk -- known at
// compile time -- and no chunk holds rows behind the end.  Chunk 0 may start above row 0: rows < 0 have a zero
// right-hand side and stay zero.
struct Row { double l1, l2, dinv; };
// rows a0-1 .. a0+M+1 inside the steady range: per-lane constants
struct L12 { double l1, l2; };
struct CoefCst {
  double l1, l2, dinv;
  __device__ __forceinline__ Row at(int) const { return {l1, l2, dinv}; }
  __device__ __forceinline__ L12 l12(int) const { return {l1, l2}; }
};
// the first rows of an utterance (not yet steady): table rows
struct CoefTop {
  __amdgpu_buffer_rsrc_t rows;  // the dim group's table rows (wave-uniform descriptor); lane offset = lane * 8
  unsigned loff;
  int i_s;
  __device__ __forceinline__ double ld(int j, int q) const { return ld_row<double>(rows, (unsigned)(4 * j + q) * 512u, loff); }
  __device__ __forceinline__ Row at(int i) const {
    int j = i < 0 ? 0 : i;
    j = j > i_s ? i_s : j;
    return {ld(j, 0), ld(j, 1), ld(j, 2)};
  }
  __device__ __forceinline__ L12 l12(int i) const {
    int j = i < 0 ? 0 : i;
    j = j > i_s ? i_s : j;
    return {ld(j, 0), ld(j, 1)};
  }
};
// rows T-2 and T-1 of an utterance from the table's state (tools/const_model.py: Coefs); true if a pivot fails
template <int NW>
__device__ __forceinline__ bool tail_rows(const CoefTop &top, int T, const double (&tau)[NW], const double (*wc)[9], Row &t0,
                                          Row &t1) {
  auto state_row = [&](int j) {  // row j < T-2 of the table (clamped), as a FacRow
    const int q = j > top.i_s ? top.i_s : j;
    return FacRow{top.ld(q, 0), top.ld(q, 1), top.ld(q, 2), top.ld(q, 3)};
  };
  const FacRow unit = {0.0, 0.0, 1.0, 1.0};
  const FacRow r3 = T - 3 >= 0 ? state_row(T - 3) : unit, r4 = T - 4 >= 0 ? state_row(T - 4) : unit;
  bool bad = false;
  FacRow ra = unit;
  double pa, pc, pe;
  if (T - 2 >= 0) {
    p_entries<NW>(T - 2, T, tau, wc, pa, pc, pe);
    ra = ldl_row(pa, pc, pe, FacState{r3.d, r3.dinv, r4.d, r4.dinv, r3.l1});
    bad |= !(ra.d > 0.0);
  }
  p_entries<NW>(T - 1, T, tau, wc, pa, pc, pe);
  const FacRow rb = ldl_row(pa, pc, pe, FacState{ra.d, ra.dinv, r3.d, r3.dinv, ra.l1});
  bad |= !(rb.d > 0.0);
  t0 = {ra.l1, ra.l2, ra.dinv};
  t1 = {rb.l1, rb.l2, rb.dinv};
  return bad;
}
// frame t above the utterance's start, or its first frame for a dynamic window: weight 0 (chunk 0 only; wave-uniform)
__device__ __forceinline__ double live_top(int w, int t) { return (t < 0 || (w != 0 && t == 0)) ? 0.0 : 1.0; }

// ---- pass 1a: the right-hand side rows of a chunk ------------------------------------------------------------------
// Forward problem: b_j = sum_w sum_t W_w[t,j] tau_w(t) mu_w(t); frame t feeds rows t-1, t, t+1.  A wavefront streams
// ITS OWN M frames through a ring of RING frames of loads (no halo frames: what its first frame adds to the row above
// and its last frame to the row below is handed to the neighbouring wavefronts through LDS -- up / dn -- and what the
// frames of the neighbouring STRIPS add comes from two extra frames read by the strip's first and last wavefront).
// Backward problem: the rows are grad_out's.  The first RING frames are issued by the caller (while the previous
// super-step is still being worked on).
// Chunk 0 may start above row 0 (weights 0 above frame 0 and for the dynamic windows on it); the last frame of the
// utterance's last chunk (= T-1) carries no dynamic precision.
constexpr int kPipe = 3;  // rows of coefficients in flight
// MULTI (several streams side by side on the lanes): the window pitch is the LANE's (its stream's static dim), so the
// window offset rides in the lane offset (wlane = pitch in bytes) instead of the scalar one (win_bytes = 0 then)
template <typename TIN, bool BWD, int NW, int RINGA, bool MULTI = false>
__device__ __forceinline__ void ring_issue(TIN (&ring)[RINGA][BWD ? 1 : NW], int slot, int k, __amdgpu_buffer_rsrc_t rs,
                                           unsigned loff, unsigned ld_bytes, unsigned win_bytes, int a0, unsigned wlane = 0u) {
  int t = a0 + k;
  t = t < 0 ? 0 : t;  // frames above the utterance's start weigh 0 (chunk 0); never read outside the utterance
#pragma unroll
  for (int w = 0; w < (BWD ? 1 : NW); ++w)
    ring[slot][w] = ld_row<TIN>(rs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, MULTI ? loff + (unsigned)w * wlane : loff);
}
template <typename TIN, bool BWD, int NW, int M, int RING, bool MULTI = false>
__device__ __forceinline__ void pass1a(TIN (&ring)[RING][BWD ? 1 : NW], __amdgpu_buffer_rsrc_t rs, unsigned loff,
                                       unsigned ld_bytes, unsigned win_bytes, int a0, bool last, const double (&tau)[NW],
                                       const double (*wc)[9], double (&z)[M], double &up, double &dn, unsigned wlane = 0u) {
  // One instruction stream for every chunk (four specialised copies behind wave-uniform branches made the register
  // allocator spill at the joins): the weights of chunk 0's dead frames are wave-uniform 0/1 factors, three scalar
  // multiplies per frame that the other chunks carry along (the pass is bound by its loads).
  const double lastm = last ? 0.0 : 1.0;  // the utterance's last frame carries no dynamic precision
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const int slot = k % RING;
    if (BWD) {
      double b = (double)ring[slot][0];
      if (a0 + k < 0) b = 0.0;
      if (k + RING < M) ring_issue<TIN, BWD, NW, RING, MULTI>(ring, slot, k + RING, rs, loff, ld_bytes, win_bytes, a0, wlane);
      z[k] = b;
    } else {
      double nx = 0.0, cu = 0.0, pv = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        double m = live_top(w, a0 + k);
        if (k == M - 1 && w != 0) m *= lastm;
        const double v = (tau[w] * m) * (double)ring[slot][w];
        nx += wc[w][2] * v;
        cu += wc[w][1] * v;
        pv += wc[w][0] * v;
      }
      if (k + RING < M) ring_issue<TIN, BWD, NW, RING, MULTI>(ring, slot, k + RING, rs, loff, ld_bytes, win_bytes, a0, wlane);
      if (k == 0) up = pv; else z[k - 1] += pv;
      if (k == 0) z[k] = cu; else z[k] += cu;
      if (k == M - 1) dn = nx; else z[k + 1] = nx;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (BWD) up = dn = 0.0;
}

// ---- pass 1b: the local forward recurrence (zero incoming state) over the rows of pass 1a ---------------------------
// TOP: the coefficients come row by row from the table (row k+kPipe's are fetched while row k is processed) and the
// chunk's transfer matrix A is formed by the homogeneous recurrence.  last: rows M-2, M-1 take the tail rows t0, t1.
// On return z[] holds the local solution, g its last two rows.
__device__ __forceinline__ Row pick(bool c, const Row &a, const Row &b) { return {c ? a.l1 : b.l1, c ? a.l2 : b.l2, c ? a.dinv : b.dinv}; }
__device__ __forceinline__ L12 pick(bool c, const Row &a, const L12 &b) { return {c ? a.l1 : b.l1, c ? a.l2 : b.l2}; }
template <int M, bool TOP, typename Coef>
__device__ __forceinline__ void pass1b(int a0, const Coef &co, bool last, const Row &t0, const Row &t1, double (&z)[M], V2 &g,
                                       M2 &A) {
  double zm1 = 0.0, zm2 = 0.0;
  double h1a = 1.0, h1b = 0.0, h2a = 0.0, h2b = 1.0;
  L12 pipe[kPipe];
#pragma unroll
  for (int j = 0; j < kPipe; ++j) pipe[j] = co.l12(a0 + j);
#pragma unroll
  for (int k = 0; k < M; ++k) {
    L12 r = pipe[k % kPipe];
    if (k + kPipe < M) pipe[k % kPipe] = co.l12(a0 + k + kPipe);
    if (k == M - 2) r = pick(last, t0, r);
    if (k == M - 1) r = pick(last, t1, r);
    const double zz = z[k] - r.l1 * zm1 - r.l2 * zm2;
    z[k] = zz;
    zm2 = zm1;
    zm1 = zz;
    if (TOP) {
      const double n1 = -r.l1 * h1a - r.l2 * h1b, n2 = -r.l1 * h2a - r.l2 * h2b;
      h1b = h1a; h1a = n1;
      h2b = h2a; h2a = n2;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  g = {zm1, zm2};
  A = {h1a, h2a, h1b, h2b};
}

// ---- pass 2: true forward state in, local backward recurrence (zero incoming from below) --------------------------
// z[] <- y-hat; e = (y-hat_{a0}, y-hat_{a0+1}); TOP: the chunk's backward transfer matrix B.
template <int M, bool TOP, typename Coef>
__device__ __forceinline__ void pass2(int a0, const Coef &co, bool last, const Row &t0, const Row &t1, const V2 &s,
                                      double (&z)[M], V2 &e, M2 &B) {
  double dm1 = s.x, dm2 = s.y;
  {
    L12 pipe[kPipe];
#pragma unroll
    for (int j = 0; j < kPipe; ++j) pipe[j] = co.l12(a0 + j);
#pragma unroll
    for (int k = 0; k < M; ++k) {
      L12 r = pipe[k % kPipe];
      if (k + kPipe < M) pipe[k % kPipe] = co.l12(a0 + k + kPipe);
      if (k == M - 2) r = pick(last, t0, r);
      if (k == M - 1) r = pick(last, t1, r);
      const double dl = -r.l1 * dm1 - r.l2 * dm2;
      z[k] += dl;
      dm2 = dm1;
      dm1 = dl;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double yp1 = 0.0, yp2 = 0.0;
  double k1a = 1.0, k1b = 0.0, k2a = 0.0, k2b = 1.0;
  // rows i+1 (its l1; its l2 one row later) and i+2 (its l2); nothing below the last chunk
  const L12 b1 = co.l12(a0 + M), b2 = co.l12(a0 + M + 1);
  double n_l1 = last ? 0.0 : b1.l1, n_l2 = last ? 0.0 : b1.l2, nn_l2 = last ? 0.0 : b2.l2;
  Row pipe[kPipe];
#pragma unroll
  for (int j = 0; j < kPipe; ++j) pipe[j] = co.at(a0 + M - 1 - j);
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {
    const int q = (M - 1 - k) % kPipe;
    Row r0 = pipe[q];
    if (k - kPipe >= 0) pipe[q] = co.at(a0 + k - kPipe);
    if (k == M - 2) r0 = pick(last, t0, r0);
    if (k == M - 1) r0 = pick(last, t1, r0);
    const double y = r0.dinv * z[k] - n_l1 * yp1 - nn_l2 * yp2;
    z[k] = y;
    yp2 = yp1;
    yp1 = y;
    if (TOP) {
      const double n1 = -n_l1 * k1a - nn_l2 * k1b, n2 = -n_l1 * k2a - nn_l2 * k2b;
      k1b = k1a; k1a = n1;
      k2b = k2a; k2a = n2;
    }
    nn_l2 = n_l2;
    n_l1 = r0.l1;
    n_l2 = r0.l2;
    __builtin_amdgcn_sched_barrier(0);
  }
  e = {yp1, yp2};
  B = {k1a, k2a, k1b, k2b};
}

// ---- pass 3: true state from below in: z[] <- y ---------------------------------------------------------------------
template <int M, typename Coef>
__device__ __forceinline__ void pass3(int a0, const Coef &co, const V2 &t, double (&z)[M]) {
  double ep1 = t.x, ep2 = t.y;
  const L12 b1 = co.l12(a0 + M), b2 = co.l12(a0 + M + 1);
  double n_l1 = b1.l1, n_l2 = b1.l2, nn_l2 = b2.l2;
  L12 pipe[kPipe];
#pragma unroll
  for (int j = 0; j < kPipe; ++j) pipe[j] = co.l12(a0 + M - 1 - j);
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {
    const int q = (M - 1 - k) % kPipe;
    const double ep = -n_l1 * ep1 - nn_l2 * ep2;
    z[k] += ep;
    ep2 = ep1;
    ep1 = ep;
    const L12 r0 = pipe[q];
    if (k - kPipe >= 0) pipe[q] = co.l12(a0 + k - kPipe);
    nn_l2 = n_l2;
    n_l1 = r0.l1;
    n_l2 = r0.l2;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------
template <int W, int M, int SLOTS>
struct Lds {
  double x[W][6][64];      // per chunk: (g, A) after pass 1b, then (e, B) after pass 2
  double halo[W][2][64];   // what a chunk's first frame adds to the row above it / its last frame to the row below
  double carry[2][3][64];  // by super-step parity: the forward state entering it (2), the row-below sum of the one before (1)
  double park[SLOTS][M][64];  // chunks that wait for the next super-step (the SLOTS lowest of a super-step)
  int bad[2];                 // lanes (systems) of the sequence that met a failing pivot
};
#ifndef MLPG_CONST_RING
#define MLPG_CONST_RING 6
#endif
constexpr int kRing = MLPG_CONST_RING;  // frames of loads in flight per wavefront

template <typename TIN, typename TOUT, bool BWD, int VM, int NW, int M, int W, int WPS, int SLOTS, bool MULTI = false>
__global__ __launch_bounds__(W * 64, WPS) void stream_kernel(Problem p, WinSet ws, Args a) {
  static_assert(!(MULTI && BWD), "merged streams: forward only");
  __shared__ Lds<W, M, SLOTS> lds;
  constexpr int kSlots = SLOTS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int S = M * W;
  constexpr int NL = BWD ? 1 : NW;
  static_assert(kRing <= M && M >= 4, "ring depth / chunk length");
  if (tid < 2) lds.bad[tid] = 0;
#ifdef MLPG_CONST_TIMING
  // cycle counts of this wavefront over all its super-steps: 0 outside the phases below (sequence set-up, the end of
  // rows out), 1 pass 1a, 2 barrier B0, 3 halo + pass 1b, 4 B1, 5 prefix + pass 2, 6 B2 + B3, 7 suffix, 8 parked rows
  // out, 9 own rows out / park; 10 verdict; 12 super-steps, 13 parked chunks, 14 sequences
  long long tq[16];
  for (int k = 0; k < 16; ++k) tq[k] = 0;
  long long t_prev = (long long)__builtin_readcyclecounter();
#endif

  // Workgroups that start together load, wait and compute in lockstep: the memory system is idle while they all compute
  // and saturated while they all load.  Every other group of 8 consecutive workgroups (the 8 XCDs take consecutive
  // workgroups in turn: every other workgroup of an XCD) starts 2 x 8128 cycles late, and since the periods are equal
  // the offset persists.  Measured (config 2 shape, global variances, float64): forward 0.143 -> 0.126 ms, backward
  // 0.140 -> 0.134; 512 x 2000 x 60: 0.460 -> 0.425.  A launch that does not fill the chip has nothing to gain (config 3
  // shape: 0.039 -> 0.045 ms with the delay), so only grids of at least half the device's CUs do it (Args::stagger, set by the launcher).
#ifndef MLPG_CONST_STAGGER
#define MLPG_CONST_STAGGER 2
#endif
#ifndef MLPG_CONST_STAGGER_SHIFT
#define MLPG_CONST_STAGGER_SHIFT 3
#endif
#ifndef MLPG_CONST_STAGGER_MASK
#define MLPG_CONST_STAGGER_MASK 1
#endif
  // (a.stagger: MLPG_CONST_STAGGER for a grid of at least half the device's CUs, 0 below -- the launcher decides from the
  // device it launches on, not from a constant tuned on one part)
  for (int q = 0; q < a.stagger * (int)((blockIdx.x >> MLPG_CONST_STAGGER_SHIFT) & MLPG_CONST_STAGGER_MASK); ++q)
    __builtin_amdgcn_s_sleep(127);
  for (int q = blockIdx.x; q < a.nsg; q += gridDim.x) {
    const int b = q / a.ndg, dg = q - b * a.ndg;
    const int Tmax = p.Tmax;
    int T = p.lengths ? p.lengths[b] : Tmax;
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    T = __builtin_amdgcn_readfirstlane(T);
    const int NC = (T + M - 1) / M;        // chunks of this utterance, aligned to its end
    const int K = (NC + W - 1) / W;        // super-steps
    const int d0 = dg * a.dgw;
    const int sd_all = MULTI ? a.sm.total : p.sd;  // MULTI: the lanes run over the static dims of all streams
    const int nd = sd_all - d0 < a.dgw ? sd_all - d0 : a.dgw;
    const bool lane_ok = lane < nd;
    const int dm = d0 + (lane_ok ? lane : nd - 1);  // idle lanes shadow the group's last dim (never stored)
    // d: the dim's output column; din: its window-0 input column; dstat: its status column; sd_l: its window pitch
    int d = dm, din = dm, dstat = dm, sd_l = p.sd;
    if (MULTI) {
      const LaneStream ls = lane_stream(a.sm, dm);
      d = ls.dout; din = ls.din; dstat = ls.dstat; sd_l = ls.sd;
    }
    const int dbase = MULTI ? 0 : d0;  // MULTI: the descriptors start at column 0 of the parent arrays
    const unsigned ldo_bytes = (unsigned)p.ld_out * (unsigned)sizeof(TOUT);
    const unsigned out_win = (unsigned)p.sd * (unsigned)sizeof(TOUT);
    const __amdgpu_buffer_rsrc_t ors = make_rsrc((TOUT *)p.out + (size_t)b * Tmax * p.ld_out + dbase);
    const unsigned ooff = (unsigned)(d - dbase) * (unsigned)sizeof(TOUT);
    const unsigned loff = (unsigned)(din - dbase) * (unsigned)sizeof(TIN);
    const __amdgpu_buffer_rsrc_t irs =
        make_rsrc(BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * p.ld_gout + dbase : (const TIN *)p.mean + (size_t)b * Tmax * p.ld_in + dbase);
    const unsigned ld_bytes = (unsigned)(BWD ? p.ld_gout : p.ld_in) * (unsigned)sizeof(TIN);
    const unsigned win_bytes = MULTI ? 0u : (unsigned)p.sd * (unsigned)sizeof(TIN);
    const unsigned wlane = MULTI ? (unsigned)sd_l * (unsigned)sizeof(TIN) : 0u;

    // this wavefront's first chunk: its frames are requested before anything else
    TIN ring[kRing][NL];
    if (wv < NC) {
      const int a0 = T - (NC - wv) * M;
#pragma unroll
      for (int f = 0; f < kRing; ++f) ring_issue<TIN, BWD, NW, kRing, MULTI>(ring, f, f, irs, loff, ld_bytes, win_bytes, a0, wlane);
    }
    // the table's head
    const int i_s_v = a.tabi[dg * 128], ok_v = a.tabi[dg * 128 + 1];
    const int kfail = a.tabi[dg * 128 + 64 + lane];
    const double *tab = a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + lane;
    double tau[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) tau[w] = tab[(hTau + w) * 64];
    const CoefCst cc = {tab[hL1 * 64], tab[hL2 * 64], tab[hDinv * 64]};
    const M2 Ainf = {tab[(hA + 0) * 64], tab[(hA + 1) * 64], tab[(hA + 2) * 64], tab[(hA + 3) * 64]};

    // padding frames: rows T .. Tmax-1 are zero-filled, a row per wavefront in turn
    if (lane_ok) {
      for (int t = T + wv; t < Tmax; t += W) {
        if (!BWD) st_row(ors, (unsigned)t * ldo_bytes, ooff, (TOUT)0);
        else
          for (int w = 0; w < NW; ++w) st_row(ors, (unsigned)t * ldo_bytes + (unsigned)w * out_win, ooff, (TOUT)0);
      }
    }
#ifndef MLPG_CONST_TIMING
    if (wv == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + dstat] = 0;
#endif
    const int i_s = __builtin_amdgcn_readfirstlane(i_s_v);
    const bool lag_ok = __builtin_amdgcn_readfirstlane(ok_v) != 0;
    const CoefTop ct = {make_rsrc(a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + kTabHead * 64), (unsigned)lane * 8u, i_s};
    bool sys_bad = kfail > 0 && kfail - 1 < T - 2;
    if (tid < 64) {  // nothing enters the first super-step
      lds.carry[0][0][tid] = 0.0;
      lds.carry[0][1][tid] = 0.0;
      lds.carry[0][2][tid] = 0.0;
    }

    // ---- rows out: the state from below in (t), the chunk's rows from `src(k)` (registers or the LDS park) -----------
    // forward problem: y_k = src(k) + eps_k; backward problem: grad[t, w*sd+d] = tau_w(t) (cm y_{t-1} + c0 y_t + cp y_{t+1})
    // (paramgen/_mlpg.py:202-281), y_{a0+M} = t.x, y_{a0-1} one more row of the backward recurrence (its z is sx, the
    // first component of the forward state that came into the chunk).  The rows leave in descending order.
    auto rows_out = [&](auto src, const auto &co, const V2 &t, const int a0c, const bool firstc, const bool lastc,
                        const double sx) __attribute__((always_inline)) {
      const bool zero_out = sys_bad;
      double ep1 = lastc ? 0.0 : t.x, ep2 = lastc ? 0.0 : t.y;
      const L12 b1 = co.l12(a0c + M), b2 = co.l12(a0c + M + 1);
      double n_l1 = b1.l1, n_l2 = b1.l2, nn_l2 = b2.l2;
      L12 pipe[kPipe];
#pragma unroll
      for (int jj = 0; jj < kPipe; ++jj) pipe[jj] = co.l12(a0c + M - 1 - jj);
      double y1 = ep1, y2 = 0.0;  // y_{k+1}, y_{k+2} (backward problem)
      auto emit = [&](int k, double ya, double yb, double yc) __attribute__((always_inline)) {  // row a0c + k
        const int tt = a0c + k;
        if (!lane_ok || (firstc && tt < 0)) return;
        if (!BWD) {
          st_row(ors, (unsigned)tt * ldo_bytes, ooff, zero_out ? (TOUT)0 : (TOUT)yb);
        } else {
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            double v = tau[w] * (a.wc[w][0] * ya + a.wc[w][1] * yb + a.wc[w][2] * yc);
            if (firstc) v *= live_top(w, tt);
            if (lastc && k == M - 1 && w != 0) v = 0.0;  // the last frame carries no dynamic precision
            st_row(ors, (unsigned)tt * ldo_bytes + (unsigned)w * out_win, ooff, zero_out ? (TOUT)0 : (TOUT)v);
          }
        }
      };
#pragma unroll
      for (int k = M - 1; k >= 0; --k) {
        const int qq = (M - 1 - k) % kPipe;
        const double ep = lastc ? 0.0 : -n_l1 * ep1 - nn_l2 * ep2;
        const double y = src(k) + ep;
        ep2 = ep1;
        ep1 = ep;
        const L12 r0 = pipe[qq];
        if (k - kPipe >= 0) pipe[qq] = co.l12(a0c + k - kPipe);
        nn_l2 = n_l2;
        n_l1 = r0.l1;
        n_l2 = r0.l2;
        if (!BWD) emit(k, 0.0, y, 0.0);
        else if (k < M - 1) emit(k + 1, y, y1, y2);
        y2 = y1;
        y1 = y;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BWD) {
        double ym = 0.0;
        if (a0c >= 1) {
          const Row rm = co.at(a0c - 1);
          const Row r0 = co.at(a0c), r1 = co.at(a0c + 1);
          ym = rm.dinv * sx - r0.l1 * y1 - r1.l2 * y2;  // (M >= 4: rows a0, a0+1 are no tail rows)
        }
        emit(0, ym, y1, y2);
      }
    };

    double z[M];
    __syncthreads();  // the carry slots are set (and the previous sequence's LDS traffic is over)
    bool redo = !lag_ok && K > 1;
    if (!redo) {
      // what a parked chunk keeps in registers
      bool parked = false, steady_p = true, first_p = false;
      V2 tl_p = {0.0, 0.0};
      M2 Bs_p = {1.0, 0.0, 0.0, 1.0};
      double sx_p = 0.0;
      int a0_p = 0;
      const int slot = W - 1 - wv;
      for (int k = 0; k < K; ++k) {
        V2 Ehat = {0.0, 0.0};
        V2 tl = {0.0, 0.0}, s_in = {0.0, 0.0};
        M2 Bs = {1.0, 0.0, 0.0, 1.0};
        const int j = k * W + wv;              // this wavefront's chunk of super-step k
        const bool dead = j >= NC;
        const int a0 = T - (NC - j) * M;       // its first row (chunk 0: possibly above row 0)
        const bool first = j == 0, last = j == NC - 1;
        const bool steady = a0 >= (i_s + 1 > 2 ? i_s + 1 : 2);  // rows a0-1 .. a0+M+1 take the steady coefficients (the tail apart)
        CST_TICK(0);
#ifdef MLPG_CONST_TIMING
        tq[12] += 1;
#endif
        // ---- pass 1a: this chunk's right-hand side rows from its own frames (requested one super-step ago)
        TIN edge[NL];  // the super-step's last wavefront: the frame below it
        const bool edge_dn = !BWD && !dead && !last && wv == W - 1;
        if (edge_dn) {
#pragma unroll
          for (int w = 0; w < NL; ++w) edge[w] = ld_row<TIN>(irs, (unsigned)(a0 + M) * ld_bytes + (unsigned)w * win_bytes, MULTI ? loff + (unsigned)w * wlane : loff);
        }
        double up = 0.0, dn = 0.0;
        if (dead) {
#pragma unroll
          for (int i = 0; i < M; ++i) z[i] = 0.0;
        } else {
          pass1a<TIN, BWD, NW, M, kRing, MULTI>(ring, irs, loff, ld_bytes, win_bytes, a0, last, tau, a.wc, z, up, dn, wlane);
        }
        // the next super-step's first frames: they travel while this one is worked on
        if (j + W < NC) {
#pragma unroll
          for (int f = 0; f < kRing; ++f) ring_issue<TIN, BWD, NW, kRing, MULTI>(ring, f, f, irs, loff, ld_bytes, win_bytes, a0 + S, wlane);
        }
        if (!BWD) {
          lds.halo[wv][0][lane] = up;
          lds.halo[wv][1][lane] = (last || dead) ? 0.0 : dn;
          if (wv == W - 1) lds.carry[(k + 1) & 1][2][lane] = (last || dead) ? 0.0 : dn;
        }
        CST_TICK(1);
        __syncthreads();  // (B0)
        CST_TICK(2);
        if (!BWD && !dead) {
          // what the neighbouring chunks' frames add to this chunk's first and last row
          double add0 = wv > 0 ? lds.halo[wv - 1][1][lane] : lds.carry[k & 1][2][lane];
          double addm = wv < W - 1 ? lds.halo[wv + 1][0][lane] : 0.0;
          if (edge_dn) {  // frame a0+M: a live frame of every window (the next chunk is full)
#pragma unroll
            for (int w = 0; w < NW; ++w) addm += a.wc[w][0] * (tau[w] * (double)edge[w]);
          }
          z[0] += add0;
          z[M - 1] += addm;
        }
        // ---- pass 1b: local forward recurrence; the chunk's hand-over (g, A)
        Row t0 = {0.0, 0.0, 1.0}, t1 = {0.0, 0.0, 1.0};
        if (last && !dead) sys_bad |= tail_rows<NW>(ct, T, tau, a.wc, t0, t1);
        V2 g2 = {0.0, 0.0};
        M2 A2 = {0.0, 0.0, 0.0, 0.0};
        if (!dead) {
          if (steady) {
            M2 dummy;
            pass1b<M, false>(a0, cc, last, t0, t1, z, g2, dummy);
            A2 = Ainf;  // (never used behind the last chunk)
          } else {
            pass1b<M, true>(a0, ct, last, t0, t1, z, g2, A2);
          }
        }
        lds.x[wv][0][lane] = g2.x; lds.x[wv][1][lane] = g2.y;
        lds.x[wv][2][lane] = A2.a; lds.x[wv][3][lane] = A2.b; lds.x[wv][4][lane] = A2.c; lds.x[wv][5][lane] = A2.d;
        CST_TICK(3);
        __syncthreads();  // (B1)
        CST_TICK(4);
        // the true state entering this chunk: the super-step's incoming state through the chunks above
        s_in = {lds.carry[k & 1][0][lane], lds.carry[k & 1][1][lane]};
        for (int c = 0; c < wv; ++c) {
          const M2 Ac = {lds.x[c][2][lane], lds.x[c][3][lane], lds.x[c][4][lane], lds.x[c][5][lane]};
          s_in = add(V2{lds.x[c][0][lane], lds.x[c][1][lane]}, mv(Ac, s_in));
        }
        if (wv == W - 1) {  // ... and the next super-step's
          const V2 sn = add(g2, mv(A2, s_in));
          lds.carry[(k + 1) & 1][0][lane] = sn.x;
          lds.carry[(k + 1) & 1][1][lane] = sn.y;
        }
        // ---- pass 2: true forward state in, local backward recurrence; the hand-over (e, B)
        // (into the exchange slots the (g, A) are read from: (B2) below)
        V2 e2 = {0.0, 0.0};
        M2 B2 = {0.0, 0.0, 0.0, 0.0};
        if (!dead) {
          if (steady) {
            M2 dummy;
            pass2<M, false>(a0, cc, last, t0, t1, s_in, z, e2, dummy);
            B2 = Ainf;  // the two recurrences share their coefficients in the steady range
          } else {
            pass2<M, true>(a0, ct, last, t0, t1, s_in, z, e2, B2);
          }
        }
        CST_TICK(5);
        __syncthreads();  // (B2) everybody has read the (g, A)
        lds.x[wv][0][lane] = e2.x; lds.x[wv][1][lane] = e2.y;
        lds.x[wv][2][lane] = B2.a; lds.x[wv][3][lane] = B2.b; lds.x[wv][4][lane] = B2.c; lds.x[wv][5][lane] = B2.d;
        __syncthreads();  // (B3)
        CST_TICK(6);
        // what comes up from the chunks below: t_c = tl_c + Bs_c T for the state T below the super-step (from the
        // bottom chunk upwards: tl_{c-1} = e_c + B_c tl_c, Bs_{c-1} = B_c Bs_c); out of the super-step's top with T = 0:
        // Ehat, what the super-step above is finished with
        {
          M2 Prun = {1.0, 0.0, 0.0, 1.0};
          for (int c = W - 1; c >= 0; --c) {
            if (c == wv) { tl = Ehat; Bs = Prun; }
            const M2 Bc = {lds.x[c][2][lane], lds.x[c][3][lane], lds.x[c][4][lane], lds.x[c][5][lane]};
            Ehat = add(V2{lds.x[c][0][lane], lds.x[c][1][lane]}, mv(Bc, Ehat));
            Prun = mm(Bc, Prun);
          }
        }
        CST_TICK(7);
        // ---- the chunk parked one super-step ago: out, with what came up from this super-step
        if (parked) {
          const V2 t_p = add(tl_p, mv(Bs_p, Ehat));
          auto from_lds = [&](int kk) __attribute__((always_inline)) { return lds.park[slot][kk][lane]; };
          if (steady_p) rows_out(from_lds, cc, t_p, a0_p, first_p, false, sx_p);
          else rows_out(from_lds, ct, t_p, a0_p, first_p, false, sx_p);
          parked = false;
        }
        CST_TICK(8);
        // ---- this chunk: out now if nothing below can reach it (the last super-step; or the chunks between it and
        // the super-step's bottom damp whatever comes up below kTol), else parked until the next super-step is through
        if (!dead) {
          const bool need = k < K - 1 && __ballot(lane_ok && !(amax4(Bs) < kTol)) != 0ull;
          if (!need) {
            auto from_reg = [&](int kk) __attribute__((always_inline)) { return z[kk]; };
            if (steady) rows_out(from_reg, cc, tl, a0, first, last, s_in.x);
            else rows_out(from_reg, ct, tl, a0, first, last, s_in.x);
          } else if (slot < kSlots) {
#pragma unroll
            for (int i = 0; i < M; ++i) lds.park[slot][i][lane] = z[i];
            parked = true;
            tl_p = tl; Bs_p = Bs; sx_p = s_in.x; a0_p = a0; steady_p = steady; first_p = first;
          } else {
            redo = true;  // no park slot this far from the bottom: the sequence takes the two-sweep path
          }
        }
      }
      // (nobody is parked after the last super-step)
      redo = __syncthreads_or(redo);
    }
    if (redo) {
      // ---- exact two-sweep path (slow decay): forward sweep, the scaled z parked in the output rows ...
      if (wv < NC) {
        const int a0 = T - (NC - wv) * M;
#pragma unroll
        for (int f = 0; f < kRing; ++f) ring_issue<TIN, BWD, NW, kRing, MULTI>(ring, f, f, irs, loff, ld_bytes, win_bytes, a0, wlane);
      }
      if (tid < 64) {
        lds.carry[0][0][tid] = 0.0;
        lds.carry[0][1][tid] = 0.0;
        lds.carry[0][2][tid] = 0.0;
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t prs = make_rsrc((TOUT *)p.out + (size_t)b * Tmax * p.ld_out + dbase);
      for (int k = 0; k < K; ++k) {
        const int j = k * W + wv;
        const bool dead = j >= NC;
        const int a0 = T - (NC - j) * M;
        const bool first = j == 0, last = j == NC - 1;
        const bool steady = a0 >= (i_s + 1 > 2 ? i_s + 1 : 2);
        TIN edge[NL];
        const bool edge_dn = !BWD && !dead && !last && wv == W - 1;
        if (edge_dn) {
#pragma unroll
          for (int w = 0; w < NL; ++w) edge[w] = ld_row<TIN>(irs, (unsigned)(a0 + M) * ld_bytes + (unsigned)w * win_bytes, MULTI ? loff + (unsigned)w * wlane : loff);
        }
        double up = 0.0, dn = 0.0;
        if (dead) {
#pragma unroll
          for (int i = 0; i < M; ++i) z[i] = 0.0;
        } else {
          pass1a<TIN, BWD, NW, M, kRing, MULTI>(ring, irs, loff, ld_bytes, win_bytes, a0, last, tau, a.wc, z, up, dn, wlane);
        }
        if (j + W < NC) {
#pragma unroll
          for (int f = 0; f < kRing; ++f) ring_issue<TIN, BWD, NW, kRing, MULTI>(ring, f, f, irs, loff, ld_bytes, win_bytes, a0 + S, wlane);
        }
        if (!BWD) {
          lds.halo[wv][0][lane] = up;
          lds.halo[wv][1][lane] = (last || dead) ? 0.0 : dn;
          if (wv == W - 1) lds.carry[(k + 1) & 1][2][lane] = (last || dead) ? 0.0 : dn;
        }
        __syncthreads();
        if (!BWD && !dead) {
          double add0 = wv > 0 ? lds.halo[wv - 1][1][lane] : lds.carry[k & 1][2][lane];
          double addm = wv < W - 1 ? lds.halo[wv + 1][0][lane] : 0.0;
          if (edge_dn) {
#pragma unroll
            for (int w = 0; w < NW; ++w) addm += a.wc[w][0] * (tau[w] * (double)edge[w]);
          }
          z[0] += add0;
          z[M - 1] += addm;
        }
        Row t0 = {0.0, 0.0, 1.0}, t1 = {0.0, 0.0, 1.0};
        if (last && !dead) sys_bad |= tail_rows<NW>(ct, T, tau, a.wc, t0, t1);
        V2 g2 = {0.0, 0.0};
        M2 A2 = {0.0, 0.0, 0.0, 0.0};
        if (!dead) {
          if (steady) {
            M2 dummy;
            pass1b<M, false>(a0, cc, last, t0, t1, z, g2, dummy);
            A2 = Ainf;
          } else {
            pass1b<M, true>(a0, ct, last, t0, t1, z, g2, A2);
          }
        }
        lds.x[wv][0][lane] = g2.x; lds.x[wv][1][lane] = g2.y;
        lds.x[wv][2][lane] = A2.a; lds.x[wv][3][lane] = A2.b; lds.x[wv][4][lane] = A2.c; lds.x[wv][5][lane] = A2.d;
        __syncthreads();
        V2 s_in = {lds.carry[k & 1][0][lane], lds.carry[k & 1][1][lane]};
        for (int c = 0; c < wv; ++c) {
          const M2 Ac = {lds.x[c][2][lane], lds.x[c][3][lane], lds.x[c][4][lane], lds.x[c][5][lane]};
          s_in = add(V2{lds.x[c][0][lane], lds.x[c][1][lane]}, mv(Ac, s_in));
        }
        if (wv == W - 1) {
          const V2 sn = add(g2, mv(A2, s_in));
          lds.carry[(k + 1) & 1][0][lane] = sn.x;
          lds.carry[(k + 1) & 1][1][lane] = sn.y;
        }
        // true z = local z + the homogeneous response to s_in; stored scaled by 1/d (what the backward recurrence takes)
        if (!dead) {
          auto park = [&](const auto &co) __attribute__((always_inline)) {
            double dm1 = s_in.x, dm2 = s_in.y;
#pragma unroll
            for (int i = 0; i < M; ++i) {
              Row r = co.at(a0 + i);
              if (i == M - 2) r = pick(last, t0, r);
              if (i == M - 1) r = pick(last, t1, r);
              const double dl = -r.l1 * dm1 - r.l2 * dm2;
              dm2 = dm1;
              dm1 = dl;
              const int t = a0 + i;
              if (lane_ok && (!first || t >= 0)) st_row(prs, (unsigned)t * ldo_bytes, ooff, (TOUT)((z[i] + dl) * r.dinv));
            }
          };
          if (steady) park(cc); else park(ct);
        }
        __syncthreads();  // the carry of super-step k has been read by everybody before k+2 overwrites its slot
      }
      // ... backward sweep, super-steps in reverse order; the state from below is carried exactly
      __syncthreads();
      if (tid < 64) { lds.carry[K & 1][0][tid] = 0.0; lds.carry[K & 1][1][tid] = 0.0; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's parked rows have left
      __syncthreads();
      for (int k = K - 1; k >= 0; --k) {
        const int j = k * W + wv;
        const bool dead = j >= NC;
        const int a0 = T - (NC - j) * M;
        const bool first = j == 0, last = j == NC - 1;
        const bool steady = a0 >= (i_s + 1 > 2 ? i_s + 1 : 2);
        Row t0 = {0.0, 0.0, 1.0}, t1 = {0.0, 0.0, 1.0};
        if (last && !dead) (void)tail_rows<NW>(ct, T, tau, a.wc, t0, t1);
        double vm1 = 0.0;  // the parked value of row a0-1 (backward problem: one more row of the recurrence)
        if (!dead) {
#pragma unroll
          for (int i = 0; i < M; ++i) {
            int t = a0 + i;
            t = t < 0 ? 0 : t;
            z[i] = (double)ld_row_fresh<TOUT>(prs, (unsigned)t * ldo_bytes, ooff);
            if (first && a0 + i < 0) z[i] = 0.0;
          }
          if (BWD && a0 >= 1) vm1 = (double)ld_row_fresh<TOUT>(prs, (unsigned)(a0 - 1) * ldo_bytes, ooff);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every parked row is in registers before anybody overwrites one
        // local backward recurrence over the parked (already scaled) values
        V2 e2 = {0.0, 0.0};
        M2 B2 = {0.0, 0.0, 0.0, 0.0};
        if (!dead) {
          auto back = [&](const auto &co) __attribute__((always_inline)) {
            double yp1 = 0.0, yp2 = 0.0, k1a = 1.0, k1b = 0.0, k2a = 0.0, k2b = 1.0;
            const L12 b1 = co.l12(a0 + M), b2 = co.l12(a0 + M + 1);
            double n_l1 = last ? 0.0 : b1.l1, n_l2 = last ? 0.0 : b1.l2, nn_l2 = last ? 0.0 : b2.l2;
#pragma unroll
            for (int i = M - 1; i >= 0; --i) {
              L12 r0 = co.l12(a0 + i);
              if (i == M - 2) r0 = pick(last, t0, r0);
              if (i == M - 1) r0 = pick(last, t1, r0);
              const double y = z[i] - n_l1 * yp1 - nn_l2 * yp2;
              z[i] = y;
              yp2 = yp1; yp1 = y;
              const double n1 = -n_l1 * k1a - nn_l2 * k1b, n2 = -n_l1 * k2a - nn_l2 * k2b;
              k1b = k1a; k1a = n1;
              k2b = k2a; k2a = n2;
              nn_l2 = n_l2; n_l1 = r0.l1; n_l2 = r0.l2;
            }
            e2 = {yp1, yp2};
            B2 = {k1a, k2a, k1b, k2b};
          };
          if (steady) back(cc); else back(ct);
        }
        lds.x[wv][0][lane] = e2.x; lds.x[wv][1][lane] = e2.y;
        lds.x[wv][2][lane] = B2.a; lds.x[wv][3][lane] = B2.b; lds.x[wv][4][lane] = B2.c; lds.x[wv][5][lane] = B2.d;
        __syncthreads();
        // the state below this chunk: the super-step's (exact, from the super-step below) through the chunks below
        V2 t_in = {lds.carry[(k + 1) & 1][0][lane], lds.carry[(k + 1) & 1][1][lane]};
        V2 top = t_in;
        for (int c = W - 1; c >= 0; --c) {
          if (c == wv) t_in = top;
          const M2 Bc = {lds.x[c][2][lane], lds.x[c][3][lane], lds.x[c][4][lane], lds.x[c][5][lane]};
          top = add(V2{lds.x[c][0][lane], lds.x[c][1][lane]}, mv(Bc, top));
        }
        if (wv == 0) { lds.carry[k & 1][0][lane] = top.x; lds.carry[k & 1][1][lane] = top.y; }
        if (!dead) {
          // (backward problem: y_{a0-1} = parked_{a0-1} - l1_{a0} y_{a0} - l2_{a0+1} y_{a0+1}: rows_out forms it from
          // sx / d_{a0-1}; hand it the parked value times d_{a0-1})
          double sx = 0.0;
          if (BWD && a0 >= 1) {
            const Row rm = steady ? cc.at(a0 - 1) : ct.at(a0 - 1);
            sx = vm1 / rm.dinv;
          }
          auto from_reg = [&](int kk) __attribute__((always_inline)) { return z[kk]; };
          if (steady) rows_out(from_reg, cc, t_in, a0, first, last, sx);
          else rows_out(from_reg, ct, t_in, a0, first, last, sx);
        }
        __syncthreads();
      }
    }
    CST_TICK(9);
    // ---- verdict: a failing pivot (the table's, above this utterance's tail, or the tail's own) gets the reference's
    // status -- the natural-order first failing pivot (linalg.pyx:79-82) -- and an all-zero column, like every other kernel
    {
      const unsigned long long m = __ballot(sys_bad && lane_ok);
      if (m != 0ull && lane == 0) {
        atomicOr(&lds.bad[0], (int)(unsigned)m);
        atomicOr(&lds.bad[1], (int)(unsigned)(m >> 32));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's rows have left
      __syncthreads();
      const unsigned long long all = (unsigned long long)(unsigned)lds.bad[0] | ((unsigned long long)(unsigned)lds.bad[1] << 32);
      if (all != 0ull && wv == 0 && lane_ok && ((all >> lane) & 1ull)) {
        int status;
        if (MULTI) {
          // the dim's own stream as a problem of its own: column slices of the parent arrays (make_view addresses column
          // w * sd + dloc of q.mean / q.var: the bases are shifted so that dloc = 0 is this dim)
          Problem q = p;
          q.sd = sd_l;
          q.D = NW * sd_l;
          q.mean = (const TIN *)p.mean + din;
          q.var = p.var ? (const void *)((const TIN *)p.var + din) : nullptr;
          const SysView<TIN, BWD> view = make_view<TIN, BWD>(q, ws, b, 0, T);
          status = first_bad_pivot<2, TIN, BWD>(view, ws);
        } else {
          const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
          status = first_bad_pivot<2, TIN, BWD>(view, ws);
        }
        if (status == 0) status = -2;
        if (p.status) p.status[(size_t)b * p.ld_status + dstat] = status;
        TOUT *out_b = (TOUT *)p.out + (size_t)b * Tmax * p.ld_out;
        for (int t = 0; t < T; ++t) {
          if (!BWD) out_b[(size_t)t * p.ld_out + d] = (TOUT)0;
          else
            for (int w = 0; w < NW; ++w) out_b[(size_t)t * p.ld_out + w * p.sd + d] = (TOUT)0;
        }
      }
      __syncthreads();
      if (tid < 2) lds.bad[tid] = 0;
    }
    CST_TICK(10);
#ifdef MLPG_CONST_TIMING
    tq[14] += 1;
#endif
  }
#ifdef MLPG_CONST_TIMING
  // profiling build only: the counters overwrite the head of the status array (tools/dbg/const_timing.py)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0 && p.status && ((int)blockIdx.x * W + wv + 1) * 16 <= p.B * p.ld_status)
    for (int k = 0; k < 16; ++k) p.status[((int)blockIdx.x * W + wv) * 16 + k] = k < 12 ? (int)(tq[k] >> 4) : (int)tq[k];
#endif
}

// ---- launcher -------------------------------------------------------------------------------------------------------
inline int resident_grid(const void *kern, int threads, int *out) {
  struct Entry { const void *kern; int dev, grid; };
  static std::mutex mu;
  static std::vector<Entry> cache;
  int dev = 0;
  MLPG_HIP_CHECK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry &e : cache)
      if (e.kern == kern && e.dev == dev) { *out = e.grid; return 0; }
  }
  int ncu = 0, per_cu = 0;
  MLPG_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  MLPG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0));
  if (ncu < 1 || per_cu < 1) {
    set_error("const kernel: occupancy query returned %d workgroups per CU on %d CUs", per_cu, ncu);
    return MLPG_HIP_ERUNTIME;
  }
  std::lock_guard<std::mutex> lk(mu);
  cache.push_back({kern, dev, ncu * per_cu});
  *out = ncu * per_cu;
  return 0;
}

constexpr int kConstW = 8;  // wavefronts (chunks) per workgroup (4, two workgroups per CU: measured, no gain -- profiles/r04_notes.md section 1)
struct Plan {
  int M, W, ndg, dgw, nsg, tab_rows;
  size_t key_off, tab_off, tabi_off, total;
};
// sd_total > 0: a merged launch (several streams side by side on the lanes): full groups of 64 lanes
inline Plan make_plan(const Problem &p, int M, int W, int sd_total = 0) {
  Plan q;
  q.M = M;
  q.W = W;
  const int sd = sd_total > 0 ? sd_total : p.sd;
  q.ndg = (sd + 63) / 64;
  q.dgw = sd_total > 0 ? 64 : (sd + q.ndg - 1) / q.ndg;
  q.nsg = p.B * q.ndg;
  q.tab_rows = p.Tmax < 4 ? 4 : p.Tmax;
  q.key_off = 0;
  q.tab_off = ((size_t)q.ndg * (kMaxWindows + 2) * 64 * sizeof(double) + 255) / 256 * 256;
  q.tabi_off = q.tab_off + (size_t)q.ndg * (kTabHead + 4 * (size_t)q.tab_rows) * 64 * sizeof(double);
  q.total = q.tabi_off + (size_t)q.ndg * 128 * sizeof(int);
  return q;
}

template <typename TIN, typename TOUT, bool BWD, int VM, int NW, int M, int W, int WPS, int SLOTS, bool MULTI = false>
int launch_cfg(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, const Plan &q, bool fresh, int device,
               unsigned long long gen, const StreamMap *smap = nullptr) {
  Args a;
  memset(&a.sm, 0, sizeof(a.sm));
  if (MULTI) a.sm = *smap;
  a.key = (double *)((char *)scratch_base + q.key_off);
  a.tab = (double *)((char *)scratch_base + q.tab_off);
  a.tabi = (int *)((char *)scratch_base + q.tabi_off);
  a.ndg = q.ndg;
  a.dgw = q.dgw;
  a.nsg = q.nsg;
  a.tab_rows = q.tab_rows;
  memset(a.wc, 0, sizeof(a.wc));
  for (int w = 0; w < ws.nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *cw = ws.c + ws.off[w];
    const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
    const double v[9] = {cm, c0, cp, cp * cp, c0 * c0, cm * cm, cp * c0, c0 * cm, cp * cm};
    for (int k = 0; k < 9; ++k) a.wc[w][k] = v[k];
  }
  auto kern = stream_kernel<TIN, TOUT, BWD, VM, NW, M, W, WPS, SLOTS, MULTI>;
  int resident = 0;
  if (int rc = resident_grid((const void *)kern, W * 64, &resident)) return rc;
  // Unit variances: the table depends on the windows and the shape only, which the HOST can compare -- the launch of
  // setup_kernel (2.7 us when it only compares its key) is skipped when this (device, stream) ran the same thing last.
  bool skip = false;
  if (VM == MLPG_HIP_VAR_UNIT && !MULTI) {
    double key[2 + 3 * kMaxWindows] = {};
    key[0] = (double)((((((long long)a.tab_rows * 1024 + a.ndg) * 128 + a.dgw) * 64 + M) * 32 + W) * 8 + NW);
    key[1] = (double)sizeof(TIN);
    for (int w = 0; w < ws.nw && w < kMaxWindows; ++w)
      for (int k = 0; k < 3; ++k) key[2 + 3 * w + k] = a.wc[w][k];
    skip = const_unit_table_cached(device, st, gen, fresh, key, 2 + 3 * kMaxWindows);
  } else {
    (void)const_unit_table_cached(device, st, gen, true, nullptr, 0);  // this launch rewrites the stream's table: forget the unit entry
  }
  if (!skip) hipLaunchKernelGGL((setup_kernel<TIN, VM, NW, MULTI>), dim3((unsigned)q.ndg), dim3(64), 0, st, p, a, M, W, fresh ? 1 : 0);
  MLPG_HIP_CHECK(hipGetLastError());
  const int grid = q.nsg < resident ? q.nsg : resident;
  // the start offset pays once the workgroups of an XCD run in lockstep, i.e. from about one workgroup per two CUs on (measured on
  // 256 CUs: 64 workgroups lose 15 %, 256 gain 12 %; profiles/r04_notes.md section 2); `resident` is CUs x workgroups per CU (1 here)
  a.stagger = 2 * grid >= resident ? MLPG_CONST_STAGGER : 0;
  note_launch(MULTI ? kCountConstMulti : kCountConst);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W * 64), 0, st, p, ws, a);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

// 16-frame chunks, 8 per super-step (128 frames): one workgroup of 8 wavefronts per CU with up to 256 registers each --
// every chunk of a super-step can be parked (the one-step lag reaches 128 frames), the whole next chunk's frames are in
// flight while this one is worked on
template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, const Plan &q, bool fresh, int device,
             unsigned long long gen) {
  auto go = [&](auto vm, auto nw) -> int {
    constexpr int VM = decltype(vm)::value, NW = decltype(nw)::value;
    // (measured on MI355X, tools/gpurun/r4_stream6.sh: 24- and 32-frame chunks are 1.4x and 2x slower -- their register
    // arrays no longer fit beside the ring)
    return launch_cfg<TIN, TOUT, BWD, VM, NW, 16, kConstW, 2, kConstW>(st, p, ws, scratch_base, q, fresh, device, gen);
  };
  using G = std::integral_constant<int, MLPG_HIP_VAR_GLOBAL>;
  using U = std::integral_constant<int, MLPG_HIP_VAR_UNIT>;
  using N2 = std::integral_constant<int, 2>;
  using N3 = std::integral_constant<int, 3>;
  if (p.var_mode == MLPG_HIP_VAR_GLOBAL) return ws.nw == 3 ? go(G{}, N3{}) : go(G{}, N2{});
  return ws.nw == 3 ? go(U{}, N3{}) : go(U{}, N2{});
}

// several streams of one batch side by side on the lanes (forward, global or unit variances, three windows)
template <typename TIN>
int launch_multi_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, const Plan &q, bool fresh, int device,
                   unsigned long long gen, const StreamMap &smap) {
  if (p.var_mode == MLPG_HIP_VAR_GLOBAL)
    return launch_cfg<TIN, TIN, false, MLPG_HIP_VAR_GLOBAL, 3, 16, kConstW, 2, kConstW, true>(st, p, ws, scratch_base, q, fresh, device, gen, &smap);
  return launch_cfg<TIN, TIN, false, MLPG_HIP_VAR_UNIT, 3, 16, kConstW, 2, kConstW, true>(st, p, ws, scratch_base, q, fresh, device, gen, &smap);
}

}  // namespace cst
}  // namespace mlpg
