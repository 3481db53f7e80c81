// Constant-coefficient MLPG kernels (algo = MLPG_HIP_ALGO_CONST; the AUTO choice for global (D,) and unit variances).
//
// With global or unit variances the precision matrix  P_d = sum_w W_w^T diag(tau_w) W_w  of a static dim depends on
// (d, T) only -- the reference tiles the (D,) variances over the frames (paramgen/_mlpg.py:169-170) and re-factorises
// the same matrix for every utterance; unit_variance_mlpg_matrix (:297-373) and autograd.UnitVarianceMLPG
// (autograd/_impl/mlpg.py:70-172) are the same matrix for every dim as well.  Here it is factorised ONCE per launch:
//
//   setup_kernel   one wavefront per dim group, lane = static dim: natural-order LDL^T of P_d for T = infinity, row by
//                  row, until the multipliers have converged to their steady state (row i_s, a few dozen frames for
//                  ordinary variances); table[i] = (l1_i, l2_i, 1/d_i, d_i), i <= i_s.
//   const_kernel   lane = static dim (every load/store moves one contiguous run of sd elements of a frame),
//                  wavefront = chunk of M frames, workgroup = strip of W chunks; persistent grid, items drawn by
//                  atomic ticket in (utterance, dim group, strip) order.  Forward and backward substitution
//                      z_i = b_i - l1_i z_{i-1} - l2_i z_{i-2},      y_i = z_i / d_i - l1_{i+1} y_{i+1} - l2_{i+2} y_{i+2}
//                  are second-order linear recurrences with data-independent coefficients: per-lane CONSTANTS in
//                  registers for every chunk inside the steady range (no reciprocal, no table), table rows for the
//                  first i_s frames, and the last two rows of an utterance (the reference zeroes the dynamic precisions
//                  on the last frame, :191-193) re-derived on the fly from the table's state.  A chunk runs the
//                  recurrences with zero incoming state and hands over (g, A): its end state for zero input and the
//                  2x2 transfer matrix of its rows -- 2 + 4 numbers per lane, through LDS inside a strip and through
//                  HBM (agent-scope stores + a flag, as the strip kernel) between strips.  The true incoming state is
//                      s = sum_k (A_{c-1} .. A_{c-k+1}) g_{c-k}
//                  summed towards the utterance's start until the product of transfer matrices is below kTol: the
//                  depth depends on the variances only (never on timing: results are bitwise repeatable), 1 strip for
//                  ordinary variances, and every strip waits only for LOCAL results of its neighbours -- no chain
//                  along the utterance.  The same upwards for the backward recurrence (e, B).  Two exchange rounds.
//   finish_kernel  reference verdict (natural-order first failing pivot, zero column) for systems whose factor met a
//                  non-positive pivot; leaves the control words zero.
// tools/const_model.py is the executable specification (tests/test_const_model.py pins it against the oracle).
// Windows: extents <= 1 with at least one dynamic window of extent 1 (mw == 1), NW = 2 or 3.
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
#ifndef MLPG_CONST_ABLATE
#define MLPG_CONST_ABLATE 0  // profiling only (wrong results): 1 no inter-workgroup waits, 2 edge chunks run the steady code
#endif

namespace mlpg {
namespace cst {

constexpr int kSpinLimit = 1 << 20;
constexpr int kCtrlLine = 32;       // ints per 128-byte line
constexpr double kTol = 1e-22;      // look-back stops once max|product of transfer matrices| is below this
constexpr int kRecD = 12;           // doubles per lane per strip record: G(2) A(4) | E(2) B(4)
constexpr int kTabHead = 16;        // doubles per lane ahead of the table rows (see setup_kernel)
#ifndef MLPG_CONST_POLL_SLEEP
#define MLPG_CONST_POLL_SLEEP 16
#endif
constexpr int kPollSleep = MLPG_CONST_POLL_SLEEP;  // x 64 cycles between two looks at a neighbour's flag

// control words (ints): line 0: [0] time-outs; line 1: ticket; line 2 + g: [0..1] mask of failing lanes, [2] time-out;
// then flags[g][Rpad]: 1 = forward record published, 2 = backward record published
__host__ __device__ inline int flag_pitch(int R) { return (R + kCtrlLine - 1) / kCtrlLine * kCtrlLine; }
__host__ __device__ inline size_t ctrl_ints(int nsg, int R) { return (size_t)(2 + nsg) * kCtrlLine + (size_t)nsg * flag_pitch(R); }
inline size_t ctrl_bytes(int nsg, int R) { return (ctrl_ints(nsg, R) * sizeof(int) + 255) / 256 * 256; }

struct Args {
  int *ctrl;
  double *rec;         // [g][R][kRecD][64]
  double *tab;         // [dim group][kTabHead + 4 * tab_rows][64]
  int *tabi;           // [dim group][128]: [0] i_s, [64 + lane] first failing row + 1 of the T = infinity factor (0: none)
  int R, ndg, dgw, nsg, tab_rows;
  // per window cm = W[t,t-1], c0 = W[t,t], cp = W[t,t+1] and the products cp cp, c0 c0, cm cm, cp c0, c0 cm, cp cm: computed
  // on the host so that the kernel holds them in scalar registers (a product of two scalar doubles formed in the kernel
  // is a vector instruction whose result stays in vector registers for the whole item loop)
  double wc[kMaxWindows][9];
};

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

__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

// Records as data-tagged granules (cdna_hip_programming.md G16, R2): every number of a record is ONE naturally aligned
// 8-byte agent-scope store and IS its own flag -- the record area holds kEmpty (all ones: a NaN no arithmetic produces;
// a value that happens to carry these very bits, an input NaN's payload, is published one bit off) until the value
// lands.  No drain, no flag word, no second round trip behind the flag.
#ifndef MLPG_CONST_GRANULES
#define MLPG_CONST_GRANULES 0
#endif
constexpr unsigned long long kEmpty = ~0ull;
__device__ __forceinline__ void st_gran(double *p, double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  u = u == kEmpty ? kEmpty - 1 : u;
  __hip_atomic_store((unsigned long long *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran(const double *p) {
  return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// six granules of this lane (p, p + 64, ...): all lanes of the wavefront wait until theirs have landed; false on time-out.
// While the record is absent only its LAST-written granule is polled (one request per lane and look).
template <int SLEEP>
__device__ __forceinline__ bool wait_gran6(const double *p, bool lane_ok, double (&v)[6], int spin_limit) {
  int spins = 0;
  for (;;) {
    const unsigned long long last = ld_gran(p + 5 * 64);
    if (__ballot(lane_ok && last == kEmpty) == 0ull) {
      unsigned long long u[6];
#pragma unroll
      for (int q = 0; q < 5; ++q) u[q] = ld_gran(p + q * 64);
      u[5] = last;
      bool all = true;
#pragma unroll
      for (int q = 0; q < 6; ++q) all &= u[q] != kEmpty;
      if (__ballot(lane_ok && !all) == 0ull) {
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = __longlong_as_double((long long)u[q]);
        return true;
      }
    }
    __builtin_amdgcn_s_sleep(SLEEP);
    if (++spins > spin_limit) return false;
  }
}

// ---- buffer loads / stores: wave-uniform descriptor, row offset in an SGPR, this lane's byte offset in one VGPR ----
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#ifndef MLPG_CONST_ST_AUX
#define MLPG_CONST_ST_AUX 0  // cache policy of the output stores (2: nt)
#endif
template <typename T>
__device__ __forceinline__ T ld_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
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
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, loff, soff, MLPG_CONST_ST_AUX);
}
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, loff, soff, MLPG_CONST_ST_AUX);
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

template <typename TIN, int VM, int NW>
__device__ __forceinline__ void lane_taus(const Problem &p, int d, double (&tau)[NW]) {
#pragma unroll
  for (int w = 0; w < NW; ++w) tau[w] = VM == MLPG_HIP_VAR_GLOBAL ? tau_of<TIN>(((const TIN *)p.var)[w * p.sd + d]) : 1.0;
}

// ---- setup: the T = infinity factor of every dim, until steady ---------------------------------------------------
// Table of one dim group, per lane: head[kTabHead] = A_inf (4), B_inf (4): transfer matrices of a chunk of M steady
// rows; l1, l2, 1/d of the steady state; tau_w (3); then rows[i] = (l1_i, l2_i, 1/d_i, d_i), i <= i_s.
enum { hA = 0, hB = 4, hL1 = 8, hL2 = 9, hDinv = 10, hTau = 11 };
template <typename TIN, int VM, int NW>
__global__ __launch_bounds__(64) void setup_kernel(Problem p, Args a, int M) {
  const int lane = threadIdx.x, dg = blockIdx.x;
  const int d0 = dg * a.dgw;
  const int nd = p.sd - d0 < a.dgw ? p.sd - d0 : a.dgw;
  const int d = d0 + (lane < nd ? lane : nd - 1);
  double tau[NW];
  lane_taus<TIN, VM, NW>(p, d, tau);
  double *tab = a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + lane;
  double *rows = tab + kTabHead * 64;
  FacState s = {1.0, 1.0, 1.0, 1.0, 0.0};
  FacRow prev = {0.0, 0.0, 1.0, 1.0};
  int kfail = 0, run = 0, i = 0;
  const int big = 0x3fffffff;
  for (; i < a.tab_rows; ++i) {
    double pa, pc, pe;
    p_entries<NW>(i, big, tau, a.wc, pa, pc, pe);
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
  if (lane == 0) a.tabi[dg * 128] = i_s;
  a.tabi[dg * 128 + 64 + lane] = kfail;
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
// Backward problem: the rows are grad_out's.  The first kPro frames of the ring are issued by the caller.
// Chunk 0 may start above row 0 (weights 0 above frame 0 and for the dynamic windows on it); the last frame of the
// utterance's last chunk (= T-1) carries no dynamic precision.
constexpr int kPro = 4;   // frames of the ring issued ahead of the class decision
constexpr int kPipe = 3;  // rows of coefficients in flight
template <typename TIN, bool BWD, int NW, int RINGA>
__device__ __forceinline__ void ring_issue(TIN (&ring)[RINGA][BWD ? 1 : NW], int slot, int k, __amdgpu_buffer_rsrc_t rs,
                                           unsigned loff, unsigned ld_bytes, unsigned win_bytes, int a0) {
  int t = a0 + k;
  t = t < 0 ? 0 : t;  // frames above the utterance's start weigh 0 (chunk 0); never read outside the utterance
#pragma unroll
  for (int w = 0; w < (BWD ? 1 : NW); ++w) ring[slot][w] = ld_row<TIN>(rs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, loff);
}
template <typename TIN, bool BWD, int NW, int M, int RING>
__device__ __forceinline__ void pass1a(TIN (&ring)[RING][BWD ? 1 : NW], __amdgpu_buffer_rsrc_t rs, unsigned loff,
                                       unsigned ld_bytes, unsigned win_bytes, int a0, bool last, const double (&tau)[NW],
                                       const double (*wc)[9], double (&z)[M], double &up, double &dn) {
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
      if (k + RING < M) ring_issue<TIN, BWD, NW, RING>(ring, slot, k + RING, rs, loff, ld_bytes, win_bytes, a0);
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
      if (k + RING < M) ring_issue<TIN, BWD, NW, RING>(ring, slot, k + RING, rs, loff, ld_bytes, win_bytes, a0);
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
template <int W>
struct Lds {
  double xch[W][6][64];  // per chunk: (g, A) after pass 1, (e, B) after pass 2
  double sin[2][64];     // state entering the strip from above / from below
  double halo[W][2][64]; // what a chunk's first frame adds to the row above it / its last frame to the row below
  int misc[16];          // [1] poll ok, [2] next ticket
};
#ifndef MLPG_CONST_RING
#define MLPG_CONST_RING 6
#endif
constexpr int kRingCst = MLPG_CONST_RING;  // frames of loads in flight per wavefront in the steady chunks
constexpr int kRingTop = 4;                // ... in the chunks that read the table
constexpr int kNTq = 24;

template <typename TIN, typename TOUT, bool BWD, int VM, int NW, int M, int W>
__global__ __launch_bounds__(W * 64, 4) void const_kernel(Problem p, WinSet ws, Args a) {
  __shared__ Lds<W> lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int S = M * W;
  constexpr int NL = BWD ? 1 : NW;
  constexpr int K0 = BWD ? 0 : -1;
  static_assert(kRingCst >= kPro && kRingCst <= M && M >= 4, "ring depth / chunk length");
  const int R = a.R;
  const long nitems = (long)a.nsg * R;
  int *ticket = a.ctrl + kCtrlLine;
#ifdef MLPG_CONST_TIMING
  // phase cycle counts of this wavefront over all its items: 0 item set-up, 1 pass 1, 2 barrier, 3 prefix / publish /
  // look-back, 4 barrier, 5 pass 2, 6 barrier, 7 suffix / publish / look-ahead, 8 barrier, 9 pass 3, 10 stores,
  // 11 ticket; 12 items, 13 edge chunks, 14 look-back steps, 15 look-ahead steps; wavefront 0: 16 publish (forward),
  // 17 waiting for the neighbours' flags, 18 their records; 19-21 the same for the backward round
  long long tq[kNTq];
  for (int k = 0; k < kNTq; ++k) tq[k] = 0;
  long long t_prev = (long long)__builtin_readcyclecounter();
#define CST_SUB(k) do { const long long t_n_ = (long long)__builtin_readcyclecounter(); tq[k] += t_n_ - t_sub; t_sub = t_n_; } while (0)
#else
#define CST_SUB(k) do {} while (0)
#endif

  // tid 0: the next item's ticket.  One returning atomic; the counter may run past nitems by one draw per workgroup.
  auto draw = [&]() __attribute__((always_inline)) { return atomicAdd(ticket, 1); };
  auto post = [&](int tk) __attribute__((always_inline)) { lds.misc[2] = tk < nitems ? tk : (int)nitems; };

  auto body = [&](const int g, const int r) __attribute__((always_inline)) {
    const int b = g / a.ndg, dg = g - b * a.ndg;
    const int Tmax = p.Tmax;
    int T = p.lengths ? p.lengths[b] : Tmax;
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    T = __builtin_amdgcn_readfirstlane(T);
    const int NC = (T + M - 1) / M;        // chunks of this utterance, aligned to its end
    const int Ract = (NC + W - 1) / W;     // strips that hold live chunks
    const int d0 = dg * a.dgw;
    const int nd = p.sd - d0 < a.dgw ? p.sd - d0 : a.dgw;
    const bool lane_ok = lane < nd;
    const int d = d0 + (lane_ok ? lane : nd - 1);  // idle lanes shadow the group's last dim (never stored)
    const int j = r * W + wv;                      // this wavefront's chunk
    const int a0 = T - (NC - j) * M;               // its first row (chunk 0: possibly above row 0)
    const unsigned ldo_bytes = (unsigned)p.ld_out * (unsigned)sizeof(TOUT);
    const unsigned out_win = (unsigned)p.sd * (unsigned)sizeof(TOUT);
    const __amdgpu_buffer_rsrc_t ors = make_rsrc((TOUT *)p.out + (size_t)b * Tmax * p.ld_out + d0);
    const unsigned ooff = (unsigned)(d - d0) * (unsigned)sizeof(TOUT);

    // padding frames: the rows >= T of the strip's static range r S .. (r+1) S - 1 are zero-filled by this wavefront's
    // share of that range (the live chunks are aligned to T, not to this grid)
    if (lane_ok && (r * W + wv + 1) * M > T) {
      for (int k = 0; k < M; ++k) {
        const int t = (r * W + wv) * M + k;
        if (t >= Tmax) break;
        if (t < T) continue;
        if (!BWD) st_row(ors, (unsigned)t * ldo_bytes, ooff, (TOUT)0);
        else
          for (int w = 0; w < NW; ++w) st_row(ors, (unsigned)t * ldo_bytes + (unsigned)w * out_win, ooff, (TOUT)0);
      }
    }
    if (r >= Ract) {  // no live chunk in this strip
#ifndef MLPG_CONST_TIMING
      if (r == 0 && wv == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + d] = 0;  // T == 0
#endif
      if (tid == 0) post(draw());
      return;
    }
    const bool xwg = (MLPG_CONST_ABLATE & 1) ? false : Ract > 1;
    const bool dead = j >= NC;

    // everything the item needs from memory is requested here, at once: the table's head, i_s, and the first frames
    const unsigned loff = (unsigned)(d - d0) * (unsigned)sizeof(TIN);
    const __amdgpu_buffer_rsrc_t irs =
        make_rsrc(BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * p.ld_gout + d0 : (const TIN *)p.mean + (size_t)b * Tmax * p.ld_in + d0);
    const unsigned ld_bytes = (unsigned)(BWD ? p.ld_gout : p.ld_in) * (unsigned)sizeof(TIN);
    const unsigned win_bytes = (unsigned)p.sd * (unsigned)sizeof(TIN);
    const int i_s_v = a.tabi[dg * 128];
    const int kfail = a.tabi[dg * 128 + 64 + lane];
    const double *tab = a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + lane;
    double tau[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) tau[w] = tab[(hTau + w) * 64];
    const double c_l1 = tab[hL1 * 64], c_l2 = tab[hL2 * 64], c_dinv = tab[hDinv * 64];
    const bool first = j == 0, last = j == NC - 1;
    TIN ring[kRingCst][NL];
    TIN edge[NL];  // the strip's first wavefront: the frame above the strip; its last: the frame below
    const bool edge_up = !BWD && !dead && wv == 0 && j > 0, edge_dn = !BWD && !dead && !last && wv == W - 1;
    if (!dead) {
#pragma unroll
      for (int q = 0; q < kPro; ++q) ring_issue<TIN, BWD, NW, kRingCst>(ring, q, q, irs, loff, ld_bytes, win_bytes, a0);
      if (!BWD && (edge_up || edge_dn)) {
        const int t = edge_up ? a0 - 1 : a0 + M;
#pragma unroll
        for (int w = 0; w < NL; ++w) edge[w] = ld_row<TIN>(irs, (unsigned)t * ld_bytes + (unsigned)w * win_bytes, loff);
      }
#pragma unroll
      for (int q = kPro; q < kRingCst; ++q) ring_issue<TIN, BWD, NW, kRingCst>(ring, q, q, irs, loff, ld_bytes, win_bytes, a0);
    }
    CST_TICK(0);
    // ---- pass 1a: this chunk's right-hand side rows from its own frames
    double z[M];
    double up = 0.0, dn = 0.0;
    if (dead) {
#pragma unroll
      for (int k = 0; k < M; ++k) z[k] = 0.0;
    } else {
      pass1a<TIN, BWD, NW, M, kRingCst>(ring, irs, loff, ld_bytes, win_bytes, a0, last, tau, a.wc, z, up, dn);
    }
    if (!BWD) {
      lds.halo[wv][0][lane] = up;
      lds.halo[wv][1][lane] = last ? 0.0 : dn;
    }
    const int i_s = __builtin_amdgcn_readfirstlane(i_s_v);
    CST_TICK(1);
    __syncthreads();  // (B0)
    CST_TICK(2);
    if (!BWD && !dead) {
      // what the neighbouring chunks' frames add to this chunk's first and last row
      double add0 = 0.0, addm = 0.0;
      if (wv > 0) add0 = lds.halo[wv - 1][1][lane];
      if (wv < W - 1) addm = lds.halo[wv + 1][0][lane];
      if (edge_up) {  // frame a0-1 (a live frame; the dynamic windows are dead on it if it is frame 0)
#pragma unroll
        for (int w = 0; w < NW; ++w) add0 += a.wc[w][2] * (tau[w] * (double)edge[w] * live_top(w, a0 - 1));
      }
      if (edge_dn) {  // frame a0+M: a live frame of every window (the next chunk is full)
#pragma unroll
        for (int w = 0; w < NW; ++w) addm += a.wc[w][0] * (tau[w] * (double)edge[w]);
      }
      z[0] += add0;
      z[M - 1] += addm;
    }

    // chunk class (wave-uniform): steady = rows a0-1 .. a0+M+1 take the steady coefficients (the tail rows apart)
    bool steady = a0 >= (i_s + 1 > 2 ? i_s + 1 : 2);
    if (MLPG_CONST_ABLATE & 6) steady = true;  // (4: no table chunks, the tail rows stay)
#ifdef MLPG_CONST_TIMING
    tq[12] += 1;
    tq[13] += (!dead && (!steady || last));
#endif
    const CoefTop ct = {make_rsrc(a.tab + (size_t)dg * (kTabHead + 4 * (size_t)a.tab_rows) * 64 + kTabHead * 64), (unsigned)lane * 8u, i_s};
    // From here on the two kinds of chunk -- steady coefficients in registers, or table rows -- run separate copies of
    // the code (`rest`), joined only when z[] is dead: values of one kind kept alive across joins for the other's later
    // use cost more registers than there are.  Both copies meet the same barriers.
    auto rest = [&](auto top_tag, const auto &co) __attribute__((always_inline)) {
    constexpr bool TOPV = decltype(top_tag)::value;
    bool sys_bad = kfail > 0 && kfail - 1 < T - 2;
    Row t0 = {0.0, 0.0, 1.0}, t1 = {0.0, 0.0, 1.0};
    if (last && !(MLPG_CONST_ABLATE & 2)) sys_bad |= tail_rows<NW>(ct, T, tau, a.wc, t0, t1);
    V2 g2 = {0.0, 0.0};
    M2 A2 = {0.0, 0.0, 0.0, 0.0};
    // ---- pass 1b: local forward recurrence
    if (!dead) {
      pass1b<M, TOPV>(a0, co, last, t0, t1, z, g2, A2);
      if (!TOPV) A2 = {tab[(hA + 0) * 64], tab[(hA + 1) * 64], tab[(hA + 2) * 64], tab[(hA + 3) * 64]};  // (unused behind the last chunk)
    }
    lds.xch[wv][0][lane] = g2.x; lds.xch[wv][1][lane] = g2.y;
    lds.xch[wv][2][lane] = A2.a; lds.xch[wv][3][lane] = A2.b; lds.xch[wv][4][lane] = A2.c; lds.xch[wv][5][lane] = A2.d;
    CST_TICK(1);
    __syncthreads();  // (B1)
    CST_TICK(2);

    // ---- forward hand-over: in-strip prefix of this chunk; wavefront 0: the strip's totals, publish, look back
    V2 sl = {0.0, 0.0};
    M2 Ap = {1.0, 0.0, 0.0, 1.0};
    auto rd_v = [&](int q) __attribute__((always_inline)) { return V2{lds.xch[q][0][lane], lds.xch[q][1][lane]}; };
    auto rd_m = [&](int q) __attribute__((always_inline)) {
      return M2{lds.xch[q][2][lane], lds.xch[q][3][lane], lds.xch[q][4][lane], lds.xch[q][5][lane]};
    };
    for (int q = 0; q < wv; ++q) {
      const M2 Aq = rd_m(q);
      sl = add(rd_v(q), mv(Aq, sl));
      Ap = mm(Aq, Ap);
    }
    double *recs = a.rec + (size_t)g * R * (kRecD * 64) + lane;   // this system group's records, this lane
    int *flags = a.ctrl + (2 + a.nsg) * kCtrlLine + (size_t)g * flag_pitch(R);
    auto wait_flag = [&](int rr, int want) __attribute__((always_inline)) {
      int spins = 0;
      for (;;) {
        int f = 0;
        if (lane == 0) f = __hip_atomic_load(flags + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ONE request per look
        if (__builtin_amdgcn_readfirstlane(f) >= want) return true;
        __builtin_amdgcn_s_sleep(kPollSleep);
        if (++spins > kSpinLimit) return false;
      }
    };
    if (wv == 0) {
#ifdef MLPG_CONST_TIMING
      long long t_sub = (long long)__builtin_readcyclecounter();
#endif
      int ok = 1;
      V2 acc = {0.0, 0.0};
      if (xwg) {
        V2 G = {0.0, 0.0};
        M2 As = {1.0, 0.0, 0.0, 1.0};
        for (int q = 0; q < W; ++q) {
          const M2 Aq = rd_m(q);
          G = add(rd_v(q), mv(Aq, G));
          As = mm(Aq, As);
        }
        double *rp = recs + (size_t)r * (kRecD * 64);
#if MLPG_CONST_GRANULES
        st_gran(rp + 0 * 64, G.x); st_gran(rp + 1 * 64, G.y);
        st_gran(rp + 2 * 64, As.a); st_gran(rp + 3 * 64, As.b); st_gran(rp + 4 * 64, As.c); st_gran(rp + 5 * 64, As.d);
#else
        st_agent(rp + 0 * 64, G.x); st_agent(rp + 1 * 64, G.y);
        st_agent(rp + 2 * 64, As.a); st_agent(rp + 3 * 64, As.b); st_agent(rp + 4 * 64, As.c); st_agent(rp + 5 * 64, As.d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        CST_SUB(16);
        if (r > 0) {
          M2 P = {1.0, 0.0, 0.0, 1.0};
          for (int k = 1; r - k >= 0; ++k) {
            const double *qp = recs + (size_t)(r - k) * (kRecD * 64);
#if MLPG_CONST_GRANULES
            double v6[6];
            if (!wait_gran6<kPollSleep>(qp, lane_ok, v6, kSpinLimit)) { ok = 0; break; }
            CST_SUB(17);
            const V2 Gk = {v6[0], v6[1]};
            const M2 Ak = {v6[2], v6[3], v6[4], v6[5]};
#else
            if (!wait_flag(r - k, 1)) { ok = 0; break; }
            CST_SUB(17);
            const V2 Gk = {ld_agent(qp + 0 * 64), ld_agent(qp + 1 * 64)};
            const M2 Ak = {ld_agent(qp + 2 * 64), ld_agent(qp + 3 * 64), ld_agent(qp + 4 * 64), ld_agent(qp + 5 * 64)};
#endif
            acc = add(acc, mv(P, Gk));
            P = mm(P, Ak);
#ifdef MLPG_CONST_TIMING
            tq[14] += 1;
#endif
            const bool done = __ballot(lane_ok && !(amax4(P) < kTol)) == 0ull;  // NaN counts as "not yet"
            CST_SUB(18);
            if (done) break;
          }
        }
      }
      lds.sin[0][lane] = acc.x;
      lds.sin[1][lane] = acc.y;
      if (lane == 0) {
        if (!ok) atomicAdd(a.ctrl, 1);
        lds.misc[1] = ok;
      }
    }
    CST_TICK(3);
    __syncthreads();  // (B2)
    CST_TICK(4);
    int timed_out = !__builtin_amdgcn_readfirstlane(lds.misc[1]);
    const V2 Sin = {lds.sin[0][lane], lds.sin[1][lane]};
    const V2 s_in = add(sl, mv(Ap, Sin));

    // ---- pass 2
    V2 e2 = {0.0, 0.0};
    M2 B2 = {0.0, 0.0, 0.0, 0.0};
    if (!dead) {
      pass2<M, TOPV>(a0, co, last, t0, t1, s_in, z, e2, B2);
      if (!TOPV) B2 = {tab[(hB + 0) * 64], tab[(hB + 1) * 64], tab[(hB + 2) * 64], tab[(hB + 3) * 64]};
    }
    lds.xch[wv][0][lane] = e2.x; lds.xch[wv][1][lane] = e2.y;
    lds.xch[wv][2][lane] = B2.a; lds.xch[wv][3][lane] = B2.b; lds.xch[wv][4][lane] = B2.c; lds.xch[wv][5][lane] = B2.d;
    CST_TICK(5);
    __syncthreads();  // (B3)
    CST_TICK(6);

    // ---- backward hand-over: in-strip suffix of this chunk; wavefront 0: totals, publish, look ahead
    V2 tl = {0.0, 0.0};
    M2 Bs = {1.0, 0.0, 0.0, 1.0};
    for (int q = W - 1; q > wv; --q) {
      const M2 Bq = rd_m(q);
      tl = add(rd_v(q), mv(Bq, tl));
      Bs = mm(Bq, Bs);
    }
    if (wv == 0) {
#ifdef MLPG_CONST_TIMING
      long long t_sub = (long long)__builtin_readcyclecounter();
#endif
      int ok = 1;
      V2 acc = {0.0, 0.0};
      if (xwg && !timed_out) {
        const M2 B0 = rd_m(0);
        const V2 E = add(rd_v(0), mv(B0, tl));
        const M2 Bst = mm(B0, Bs);
        double *rp = recs + (size_t)r * (kRecD * 64);
#if MLPG_CONST_GRANULES
        st_gran(rp + 6 * 64, E.x); st_gran(rp + 7 * 64, E.y);
        st_gran(rp + 8 * 64, Bst.a); st_gran(rp + 9 * 64, Bst.b); st_gran(rp + 10 * 64, Bst.c); st_gran(rp + 11 * 64, Bst.d);
#else
        st_agent(rp + 6 * 64, E.x); st_agent(rp + 7 * 64, E.y);
        st_agent(rp + 8 * 64, Bst.a); st_agent(rp + 9 * 64, Bst.b); st_agent(rp + 10 * 64, Bst.c); st_agent(rp + 11 * 64, Bst.d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + r, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        CST_SUB(19);
        if (r + 1 < Ract) {
          M2 P = {1.0, 0.0, 0.0, 1.0};
          for (int k = 1; r + k < Ract; ++k) {
            const double *qp = recs + (size_t)(r + k) * (kRecD * 64);
#if MLPG_CONST_GRANULES
            double v6[6];
            if (!wait_gran6<kPollSleep>(qp + 6 * 64, lane_ok, v6, kSpinLimit)) { ok = 0; break; }
            CST_SUB(20);
            const V2 Ek = {v6[0], v6[1]};
            const M2 Bk = {v6[2], v6[3], v6[4], v6[5]};
#else
            if (!wait_flag(r + k, 2)) { ok = 0; break; }
            CST_SUB(20);
            const V2 Ek = {ld_agent(qp + 6 * 64), ld_agent(qp + 7 * 64)};
            const M2 Bk = {ld_agent(qp + 8 * 64), ld_agent(qp + 9 * 64), ld_agent(qp + 10 * 64), ld_agent(qp + 11 * 64)};
#endif
            acc = add(acc, mv(P, Ek));
            P = mm(P, Bk);
#ifdef MLPG_CONST_TIMING
            tq[15] += 1;
#endif
            const bool done = __ballot(lane_ok && !(amax4(P) < kTol)) == 0ull;
            CST_SUB(21);
            if (done) break;
          }
        }
      }
      lds.sin[0][lane] = acc.x;
      lds.sin[1][lane] = acc.y;
      if (lane == 0) {
        if (!ok) atomicAdd(a.ctrl, 1);
        lds.misc[1] = ok && !timed_out;
      }
    }
    // the next item's ticket: drawn now that this item has no wait left (drawn earlier, a workgroup could hold the very
    // strip its current item waits for); the atomic's round trip runs under pass 3 and the stores, its result is
    // posted at the end of the item
    int next_tk = 0;
    if (tid == 0) next_tk = draw();
    CST_TICK(7);
    __syncthreads();  // (B4)
    CST_TICK(8);
    timed_out = !__builtin_amdgcn_readfirstlane(lds.misc[1]);
    const V2 Tin = {lds.sin[0][lane], lds.sin[1][lane]};
    const V2 t_in = add(tl, mv(Bs, Tin));

    // ---- pass 3 (nothing comes in from below the last chunk)
    if (!dead && !last) pass3<M>(a0, co, t_in, z);

    CST_TICK(9);
    // ---- verdict marks: a failing pivot (the table's, below this utterance's tail, or the tail's own) or a time-out
    {
      const unsigned long long m = timed_out ? ~0ull : __ballot(sys_bad && lane_ok);
      if (m != 0ull && lane == 0) {
        int *line = a.ctrl + (2 + g) * kCtrlLine;
        __hip_atomic_fetch_or(line + 0, (int)(unsigned)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(line + 1, (int)(unsigned)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (timed_out) __hip_atomic_store(line + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#ifndef MLPG_CONST_TIMING
      if (r == 0 && wv == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + d] = 0;
#endif
    }
    const bool zero_out = sys_bad || timed_out;
    if (lane_ok && !dead) {
    const bool top = first;  // the chunk may start above row 0
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int t = a0 + k;
        if (!top || t >= 0) st_row(ors, (unsigned)t * ldo_bytes, ooff, zero_out ? (TOUT)0 : (TOUT)z[k]);
      }
    } else {
      // grad[t, w*sd+d] = tau_w(t) (cm y_{t-1} + c0 y_t + cp y_{t+1})   (paramgen/_mlpg.py:202-281); y_{a0+M} is the
      // state that came in from below (0 below the last chunk), y_{a0-1} one more row of the backward recurrence (its z
      // is the state that came in from above)
      double ym = 0.0;
      if (a0 >= 1) {
        auto three = [&](const auto &co) __attribute__((always_inline)) {
          const Row rm = co.at(a0 - 1);
          Row r0 = co.at(a0), r1 = co.at(a0 + 1);
          return rm.dinv * s_in.x - r0.l1 * z[0] - r1.l2 * z[1];  // (M >= 4: rows a0, a0+1 are no tail rows)
        };
        ym = three(co);
      }
      const double yp = last ? 0.0 : t_in.x;
      const bool lastc = last;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int t = a0 + k;
        const double ya = k == 0 ? ym : z[k - 1], yb = z[k], yc = k == M - 1 ? yp : z[k + 1];
        if (!top || t >= 0) {
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            double v = tau[w] * (a.wc[w][0] * ya + a.wc[w][1] * yb + a.wc[w][2] * yc);
            if (top) v *= live_top(w, t);
            if (lastc && k == M - 1 && w != 0) v = 0.0;  // the last frame carries no dynamic precision
            st_row(ors, (unsigned)t * ldo_bytes + (unsigned)w * out_win, ooff, zero_out ? (TOUT)0 : (TOUT)v);
          }
        }
      }
    }
    }
    if (tid == 0) post(next_tk);
    };  // rest
    if (MLPG_CONST_ABLATE & 8) rest(std::true_type{}, ct);  // (experiment: table rows everywhere)
    else if (steady) rest(std::false_type{}, CoefCst{c_l1, c_l2, c_dinv});
    else rest(std::true_type{}, ct);
  };  // body

  if (tid == 0) post(draw());
  for (;;) {
    __syncthreads();
    const int tk = __builtin_amdgcn_readfirstlane(lds.misc[2]);
    if (tk >= nitems) break;
    CST_TICK(11);
    const int g = tk / R, r = tk - g * R;
    __syncthreads();  // everybody has read the ticket: the body may overwrite it
    body(g, r);
#ifdef MLPG_CONST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CST_TICK(10);
  }
#ifdef MLPG_CONST_TIMING
  // profiling build only: the counters overwrite the head of the status array (tools/dbg/const_timing.py)
  if (lane == 0 && p.status && ((int)blockIdx.x * W + wv + 1) * kNTq <= p.B * p.ld_status)
    for (int k = 0; k < kNTq; ++k) p.status[((int)blockIdx.x * W + wv) * kNTq + k] = (k < 12 || k >= 16) ? (int)(tq[k] >> 4) : (int)tq[k];
#endif
}

// ---- finish: the reference's verdict for marked systems; leaves the control words zero ---------------------------
template <typename TIN, typename TOUT, bool BWD>
__global__ void __launch_bounds__(256) finish_kernel(const Problem p, const WinSet ws, const Args a) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= a.nsg) return;
  int *line = a.ctrl + (2 + g) * kCtrlLine;
  const unsigned long long m = (unsigned long long)(unsigned)line[0] | ((unsigned long long)(unsigned)line[1] << 32);
  const int timed_out = line[2];
  __builtin_amdgcn_wave_barrier();
  if (lane < 3) line[lane] = 0;
  for (int i = lane; i < flag_pitch(a.R); i += 64) a.ctrl[(2 + a.nsg) * kCtrlLine + (size_t)g * flag_pitch(a.R) + i] = 0;
  if (g == 0)
    for (int i = lane; i < 2 * kCtrlLine; i += 64) a.ctrl[i] = 0;
  if (m == 0ull) return;
  const int b = g / a.ndg, dg = g - b * a.ndg;
  const int d0 = dg * a.dgw;
  const int nd = p.sd - d0 < a.dgw ? p.sd - d0 : a.dgw;
  if (lane >= nd || !((m >> lane) & 1ull)) return;
  const int d = d0 + lane;
  const int Tmax = p.Tmax;
  int T = p.lengths ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  int status = -1;
  if (!timed_out) {
    const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
    status = first_bad_pivot<2, TIN, BWD>(view, ws);
    if (status == 0) status = -2;
  }
  if (p.status) p.status[(size_t)b * p.ld_status + d] = status;
  TOUT *out_b = (TOUT *)p.out + (size_t)b * Tmax * p.ld_out;
  for (int t = 0; t < Tmax; ++t) {
    if (!BWD) out_b[(size_t)t * p.ld_out + d] = (TOUT)0;
    else
      for (int w = 0; w < ws.nw; ++w) out_b[(size_t)t * p.ld_out + w * p.sd + d] = (TOUT)0;
  }
}

// ---- launcher -------------------------------------------------------------------------------------------------------
constexpr int kNotResident = -1000;  // the grid cannot hold an utterance's strips; nothing was enqueued

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

struct Plan {
  int M, W, R, ndg, dgw, nsg, tab_rows;
  size_t ctrl_off, rec_off, tab_off, tabi_off, total;
};
inline Plan make_plan(const Problem &p, int M, int W) {
  Plan q;
  q.M = M;
  q.W = W;
  q.R = (p.Tmax + M * W - 1) / (M * W);
  q.ndg = (p.sd + 63) / 64;
  q.dgw = (p.sd + q.ndg - 1) / q.ndg;
  q.nsg = p.B * q.ndg;
  q.tab_rows = p.Tmax < 4 ? 4 : p.Tmax;
  q.ctrl_off = 0;
  q.rec_off = ctrl_bytes(q.nsg, q.R);
  q.tab_off = q.rec_off + (size_t)q.nsg * q.R * kRecD * 64 * sizeof(double);
  q.tabi_off = q.tab_off + (size_t)q.ndg * (kTabHead + 4 * (size_t)q.tab_rows) * 64 * sizeof(double);
  q.total = q.tabi_off + (size_t)q.ndg * 128 * sizeof(int);
  return q;
}

template <typename TIN, typename TOUT, bool BWD, int VM, int NW, int M, int W>
int launch_cfg(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, const Plan &q, bool zero_ctrl) {
  Args a;
  a.ctrl = (int *)((char *)scratch_base + q.ctrl_off);
  a.rec = (double *)((char *)scratch_base + q.rec_off);
  a.tab = (double *)((char *)scratch_base + q.tab_off);
  a.tabi = (int *)((char *)scratch_base + q.tabi_off);
  a.R = q.R;
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
    for (int q = 0; q < 9; ++q) a.wc[w][q] = v[q];
  }
  auto kern = const_kernel<TIN, TOUT, BWD, VM, NW, M, W>;
  int resident = 0;
  if (int rc = resident_grid((const void *)kern, W * 64, &resident)) return rc;
  const long nitems = (long)q.nsg * q.R;
  if (nitems > resident && q.R > resident) return kNotResident;
  if (zero_ctrl) MLPG_HIP_CHECK(hipMemsetAsync(a.ctrl, 0, ctrl_ints(q.nsg, q.R) * sizeof(int), st));
#if MLPG_CONST_GRANULES
  // every granule of the records starts out empty (all ones)
  if (q.R > 1) MLPG_HIP_CHECK(hipMemsetAsync(a.rec, 0xFF, (size_t)q.nsg * q.R * kRecD * 64 * sizeof(double), st));
#endif
  hipLaunchKernelGGL((setup_kernel<TIN, VM, NW>), dim3((unsigned)q.ndg), dim3(64), 0, st, p, a, M);
  MLPG_HIP_CHECK(hipGetLastError());
  const long grid = nitems < resident ? nitems : resident;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W * 64), 0, st, p, ws, a);
  MLPG_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL((finish_kernel<TIN, TOUT, BWD>), dim3((unsigned)((q.nsg + 3) / 4)), dim3(256), 0, st, p, ws, a);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

// shape: 0 = large launches (32-frame chunks, 4 per strip), 1 = small launches (16-frame chunks, 2 per strip)
template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, const Plan &q, bool zero_ctrl) {
  auto go = [&](auto vm, auto nw) -> int {
    constexpr int VM = decltype(vm)::value, NW = decltype(nw)::value;
    if (q.M == 32 && q.W == 4) return launch_cfg<TIN, TOUT, BWD, VM, NW, 32, 4>(st, p, ws, scratch_base, q, zero_ctrl);
    if (q.M == 16 && q.W == 8) return launch_cfg<TIN, TOUT, BWD, VM, NW, 16, 8>(st, p, ws, scratch_base, q, zero_ctrl);
    if (q.M == 16 && q.W == 4) return launch_cfg<TIN, TOUT, BWD, VM, NW, 16, 4>(st, p, ws, scratch_base, q, zero_ctrl);
    return launch_cfg<TIN, TOUT, BWD, VM, NW, 16, 2>(st, p, ws, scratch_base, q, zero_ctrl);
  };
  using G = std::integral_constant<int, MLPG_HIP_VAR_GLOBAL>;
  using U = std::integral_constant<int, MLPG_HIP_VAR_UNIT>;
  using N2 = std::integral_constant<int, 2>;
  using N3 = std::integral_constant<int, 3>;
  if (p.var_mode == MLPG_HIP_VAR_GLOBAL) return ws.nw == 3 ? go(G{}, N3{}) : go(G{}, N2{});
  return ws.nw == 3 ? go(U{}, N3{}) : go(U{}, N2{});
}

}  // namespace cst
}  // namespace mlpg
