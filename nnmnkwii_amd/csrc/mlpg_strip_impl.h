// Strip MLPG kernels (algo = MLPG_HIP_ALGO_STRIP; the AUTO choice for wide streams).
//
// Mapping (the transpose of the wave-per-system kernel in mlpg_wave_impl.h):
//   lane      = static dim d        (adjacent lanes = adjacent columns of the row-major input:
//                                    every load/store instruction moves one contiguous run of
//                                    sd elements of a frame, no LDS transposition)
//   wavefront = chunk of M = 16 consecutive frames of one utterance
//   workgroup = strip of W = 4 consecutive chunks (64 frames); two workgroups per CU
//   an utterance of T frames is ceil(T / 64) strips that run on different CUs.
// Windows must have extents l, u <= 1 (pentadiagonal P), any T.
//
// The solve is a three-level substructured LDL^T (tools/strip_model.py is the executable
// specification, pinned against the oracle by tests/test_strip_model.py):
//   level 1 (wavefront, registers): assemble the chunk's rows of P = sum_w W_w^T diag(tau_w) W_w
//     and b from the frames f0-1 .. f0+16 -- streamed in frame order with a ring of 6 frames of loads
//     in flight (assemble_eliminate; window-major assemble + eliminate for window sets other than
//     three windows) -- and eliminate the 14 interior frames as their rows complete, carrying the
//     two "left spike" columns that couple the chunk to the previous chunk's last two frames (its
//     separator); the elimination runs on into the chunk's own separator -> 14 numbers per lane;
//   level 2 (workgroup, LDS): the block-tridiagonal system (2x2 blocks) of the strip's W
//     separators is eliminated sequentially, lanes = dims in lockstep, carrying the spike block
//     that couples the strip to the previous strip's last separator -> one 14-number record;
//   level 3 (utterance, HBM): the strips publish their records (agent-scope write-through
//     stores, a flag per strip, an arrival counter per utterance).  A strip first solves over the
//     records of strips r-2 .. r+2 only, with a rigorous bound on what the strips outside that
//     window could contribute; where the bound is not far below rounding (or the strip's own
//     transfer factor says so beforehand) it waits for the whole utterance and sweeps all records.
//     Either way a two-sided block sweep (top-down, bottom-up, 2-block system in the middle) over
//     records staged through LDS by wavefronts 1-3; no factor of the sweep is stored;
//   back-substitution in the reverse order; trajectory rows stored straight from registers.
// Edge rules (frames >= T, zeroed dynamic precisions on the first/last mw frames) are
// wave-uniform in this mapping: clamped load rows and 0/1 factors, no branches.
//
// Inter-workgroup protocol (cdna_hip_programming.md G16, placement independent): a persistent
// grid of resident workgroups draws (dim group, strip) items from per-XCD atomic ticket lists in
// (utterance, strip) order, so the strips a workgroup waits for are always held by running
// workgroups (dispatch order is not assumed); record words are stored and loaded with agent scope
// (sc1); flags and counters are relaxed agent-scope atomics polled by one wavefront with bounded
// spins; the control words are zero at launch (memset, or left so by verdict_kernel).  Systems
// with a failing pivot are marked per utterance and settled by verdict_kernel behind the main kernel.
//
// Reference semantics as in mlpg_wave_impl.h (paramgen/_mlpg.py:92-199, :202-281).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include <mutex>
#include <vector>
#include "assemble.h"

#ifndef MLPG_STRIP_ABLATE
#define MLPG_STRIP_ABLATE 0  // profiling only: 1 no inter-workgroup level, 2 = 1 + no elimination, 3 = 2 + no assembly arithmetic
#endif
#ifdef MLPG_STRIP_TIMING
#define STRIP_TICK(k)                                                  \
  do {                                                                 \
    const long long t_now_ = (long long)__builtin_readcyclecounter(); \
    tq[k] += t_now_ - t_prev;                                          \
    t_prev = t_now_;                                                   \
  } while (0)
#else
#define STRIP_TICK(k) do {} while (0)
#endif

namespace mlpg {
namespace strip {

#ifndef MLPG_STRIP_W
#define MLPG_STRIP_W 4   // 8: 128-frame strips, one workgroup of 8 wavefronts per CU (experiment; mlpg_strip.hip must match)
#endif
#ifndef MLPG_STRIP_M
#define MLPG_STRIP_M 16  // 8: the 8-frame-chunk experiment of round 5 (profiles/r05_notes.md); mlpg_strip.hip must match
#endif
#ifndef MLPG_STRIP_WGS
#define MLPG_STRIP_WGS (MLPG_STRIP_W <= 4 ? 2 : 1)  // workgroups per CU the kernel is compiled for (register budget 512 / (W/4 * WGS))
#endif
constexpr int kW = MLPG_STRIP_W;  // chunks (wavefronts) per strip (workgroup)
constexpr int kM = MLPG_STRIP_M;  // frames per chunk
constexpr int kN = kM - 2;   // interior frames of a chunk; frames kN, kN+1 are its separator
constexpr int kRec = 14;     // doubles per lane in a level-1 / level-2 record
#ifndef MLPG_STRIP_DAMP1_TOL
#define MLPG_STRIP_DAMP1_TOL 0x1p-66  // acceptance bound of the 3-strip window (route 1), see kDamp1Tol
#endif
#ifndef MLPG_STRIP_STAGE
#define MLPG_STRIP_STAGE (MLPG_STRIP_W > 6 ? 8 : 6)
#endif
constexpr int kStage = MLPG_STRIP_STAGE;  // records of level 3 staged in LDS per batch (even, >= kW)
constexpr bool kParkOn = kM >= 16;  // (an 8-frame chunk's factor stays in registers)
constexpr int kPark = kParkOn ? 2 * kN : 0;  // doubles per lane that wavefront 0 parks in LDS while it runs levels 2 and 3
constexpr int kFac = 10;     // doubles per lane kept per eliminated separator of level 2
constexpr int kSpinLimit = 1 << 20;

// record slots
enum { rT00, rT01, rT11, rH0, rH1, rD11, rD12, rD22, rF1, rF2, rL11, rL12, rL21, rL22 };

// Control words, one per 128-byte line (32 ints) so that the pollers of one utterance, the ticket draws and the
// arrivals of other utterances never queue on the same L2 line:
//   line 0: spin time-outs;  lines 1 .. 8: ticket of work list x;  line 9 + g: system group g -- word 0 arrivals,
//   words 2-3 mask of the lanes (systems) that met a failing pivot, word 4 time-out seen;
//   then one flag per strip, Rpad = R rounded up to a line per system group: flag[g * Rpad + r].
constexpr int kCtrlLine = 32;
#ifndef MLPG_STRIP_POLL_SLEEP
#define MLPG_STRIP_POLL_SLEEP 16  // x 64 cycles between two looks at the neighbours' flags (4 .. 64 measured: no difference)
#endif
#ifndef MLPG_STRIP_RING_F32
#define MLPG_STRIP_RING_F32 6
#endif
#ifndef MLPG_STRIP_BWD_FRAME_MAJOR
#define MLPG_STRIP_BWD_FRAME_MAJOR 1  // backward epilogue frame by frame (three adjacent row stores, ring of variance loads)
#endif
#ifndef MLPG_STRIP_BWD_KEEP
#define MLPG_STRIP_BWD_KEEP 1  // backward, float32 inputs, per-frame variances: the 51 precisions of a chunk (float32 values: 1/var is
                               // formed in float32, _mlpg.py:188) stay in registers from the assembly to the epilogue -- no second
                               // pass over the variances, no second division (round 5)
#endif
#ifndef MLPG_STRIP_FWD_BUFSTORE
#define MLPG_STRIP_FWD_BUFSTORE 1  // forward: trajectory rows by buffer stores (three interleaved rounds, profiles/r05_strip_ab3.txt: float64
                                   // 0.2323 / 0.2320 / 0.2376 -> 0.2281 / 0.2300 / 0.2330 ms; float32 within the noise)
#endif
#ifndef MLPG_STRIP_BWD_BUFSTORE
#define MLPG_STRIP_BWD_BUFSTORE 1  // backward epilogue (three windows): gradient rows by buffer stores (scalar offsets) instead of global stores
#endif
#ifndef MLPG_STRIP_BWD_KARG
#define MLPG_STRIP_BWD_KARG 1  // backward, three windows: window coefficients by placed scalar loads from the argument segment (karg_f64x6)
#endif
#ifndef MLPG_STRIP_FWD_KARG
#define MLPG_STRIP_FWD_KARG 1  // forward, three windows: the same for all nine coefficients per window (interleaved A/B, profiles/r05_strip_ab2.txt:
                               // float64 0.2421 / 0.2387 -> 0.2415 / 0.2352 ms, float32 0.1830 / 0.1774 -> 0.1793 / 0.1762: 146 -> 119 spilled SGPRs)
#endif
#ifndef MLPG_STRIP_BWD_KEEP0
#define MLPG_STRIP_BWD_KEEP0 1  // ... wavefront 0 too (it runs levels 2 and 3 meanwhile and has no 51 registers to spare: it reads
                                // its chunk's variances again)
#endif
#ifndef MLPG_STRIP_BWD_EARLY0
#define MLPG_STRIP_BWD_EARLY0 0  // MLPG_STRIP_BWD_EARLY for wavefront 0 (requested behind its publish)
#endif
#ifndef MLPG_STRIP_BWD_EARLY
#define MLPG_STRIP_BWD_EARLY 4  // backward, float64 in and out: the variance rows of the epilogue's first n frames are requested as soon as
                                // level 1 is done (into the registers the ring has left), so that they travel while the strip waits for
                                // its neighbours; the other 17 - n frames are requested in front of the first store as before (round 5).
                                // Worth half a percent (0 / 8 / 10 rows: 0.2671 / 0.2652 / 0.2646 ms); 4 is what fits without a spill
                                // beside the level-3 ladder's state (8: 15 spilled registers)
#endif
#ifndef MLPG_STRIP_NT_STORES
#define MLPG_STRIP_NT_STORES 1  // trajectory and gradient rows with the nontemporal hint (written once, never read by this kernel); round 5 A/B
                                // (profiles/r05_strip_stores_ab.txt): forward f64 +1.9 %, backward f32 +5.5 %.  (Round 3 measured the same gain and
                                // its notes called it shipped, but the default stayed 0 until round 5.)
#endif
#ifndef MLPG_STRIP_RING_F64
#define MLPG_STRIP_RING_F64 6
#endif
#ifndef MLPG_STRIP_STAGGER_DEFAULT_US
#define MLPG_STRIP_STAGGER_DEFAULT_US 0  // span of the start ramp in microseconds (see launch_impl); MLPG_STRIP_STAGGER_US overrides it at run time.
                                         // Off: 20 us is worth -2.4 % on config 2's forward pass (variances of one order of magnitude: the 3-strip
                                         // window) but costs +3 % where the strips wait for whole utterances (dynamic variances 100 x tighter, the
                                         // usual case for unnormalised acoustic features) and +3 % on the float32 backward (profiles/r06_notes.md)
#endif

#ifndef MLPG_STRIP_ROUTE1_TOL
#define MLPG_STRIP_ROUTE1_TOL 0x1p-66  // own transfer factor below which the 3-strip window is tried first (0: never), see kDamp1Tol
#endif
constexpr int kMaxLists = 8;
constexpr int kLocal = 2;          // level 3 first looks at the records of strips r-2 .. r+2 only
                                   // (wider windows -- 4, 8, 16 strips per side -- when the strip's own transfer factor calls for them)
constexpr double kRouteTol = 1e-11; // a strip whose own transfer factor 2 max|E^-1 V| exceeds this (= kDampTol^(1/2)) does not
                                    // try the 5-strip window (see the route)
constexpr double kDampTol = 1e-22; // ... and accepts that if the window's edges are damped below this at rows r-1, r
// The 3-strip window r-1 .. r+1 (round 5).  Separators 64 frames apart are coupled by 1e-24 .. 1e-21 for variances of one order of
// magnitude, so the rigorous bound of ONE strip per side lands around 1e-22 -- on either side of kDampTol (tools/strip_model.py on
// the bench data: median 1.0e-22, maximum 1.1e-21, while the windowed result equalled the exact solve to the last bit in every
// strip).  Its acceptance bound is therefore 2^-66 = 1.4e-20: what the window leaves out moves the separator values by at most
// 2^-13 of the spacing of doubles there.  A strip tries it when its own transfer factor is below that; a rejected attempt is followed
// by the 5-strip window, then by the whole utterance (the ladder in the kernel).  Measured: config 2 forward 0.233 -> 0.216-0.225 ms
// (profiles/r05_notes.md section 11): the strip waits for two neighbours instead of four and sweeps three records instead of five.
constexpr double kDamp1Tol = MLPG_STRIP_DAMP1_TOL;
// the records strip r of Ract reads first: rows lo .. hiE; the last one only as the clamped edge (T, h, V) if `edge`
struct Window { int lo, hiE, edge; };
__device__ __forceinline__ Window local_window(int r, int Ract, int k) {
  Window w;
  w.lo = r - k < 0 ? 0 : r - k;
  w.edge = r + k < Ract - 1;
  w.hiE = w.edge ? r + k : Ract - 1;
  return w;
}
// Staged order of the records of rows lo .. hiE around strip r: pairs (top row, bottom row) moving inwards from both
// ends while both sides have rows -- the two eliminations are independent chains, a pair is handled in one
// straight-line block so that their latencies overlap -- then the rest of the longer side, row r last.
struct Order { int lo, hiE, r, nt, nb, m, npos; };
__device__ __forceinline__ Order make_order(int r, int lo, int hiE) {
  Order o;
  o.lo = lo; o.hiE = hiE; o.r = r;
  o.nt = r - lo;    // top rows lo .. r-1
  o.nb = hiE - r;   // bottom rows hiE .. r+1
  o.m = o.nt < o.nb ? o.nt : o.nb;
  o.npos = o.nt + o.nb + 1;
  return o;
}
__device__ __forceinline__ int row_of(const Order &o, int pos) {
  if (pos < 2 * o.m) return (pos & 1) ? o.hiE - (pos >> 1) : o.lo + (pos >> 1);
  if (pos < o.nt + o.nb) {
    const int k = o.m + (pos - 2 * o.m);
    return o.nt > o.nb ? o.lo + k : o.hiE - k;
  }
  return o.r;
}
__host__ __device__ inline int flag_pitch(int R) { return (R + kCtrlLine - 1) / kCtrlLine * kCtrlLine; }
__host__ __device__ inline size_t ctrl_ints(int nsg, int R) {
  return (size_t)(1 + kMaxLists + nsg) * kCtrlLine + (size_t)nsg * flag_pitch(R);
}

struct Args {
  int *ctrl;
  double *rec;   // [g][R][kRec][64]
  int R;         // strips per system group (from Tmax)
  int ndg, dgw;  // dim groups per utterance, dims per group (<= 64)
  int nsg;       // system groups: B * ndg
  double one;                 // 1.0, from the host: the unit-variance precision as a run-time value (see assemble_eliminate)
  double wc[kMaxWindows][9];  // per window: W[t,t-1], W[t,t], W[t,t+1] and their six products (host-computed, so
                              // that the kernel holds them in scalar registers)
  int nlists;    // work lists: 8 (drawn first by the workgroups that run on XCD list % 8, so that neighbouring strips share
                 // one L2) or 1 (small launches)
  int nb, bs;    // an utterance's R strips are dealt to the lists in nb blocks of bs consecutive strips (nb * bs >= R;
                 // block j of system group g is "virtual group" g * nb + j, virtual group v belongs to list v % nlists):
                 // nb = 1, bs = R while an utterance fits into a list's share of the grid, more blocks for long ones
  StreamMap sm;  // MULTI kernels only: the streams whose static dims sit side by side on the lanes
  int stagger;   // start ramp: the workgroup that draws ticket tk < wpl of a list as its FIRST item starts it tk * stagger / wpl ticks
  int wpl;       // (100 MHz) late; wpl = workgroups per list.  0: everybody starts at once.  See launch_impl.
};

// Ticket tk of work list lst -> (system group, strip).  The lists hold whole blocks of consecutive strips in the order of
// the virtual groups, i.e. every list runs through the utterances in the same order (see launch(): what the route-0 wait
// relies on).  MULTI: the lists are dealt by UTTERANCE (all its dim groups in one list, a full group and the narrow last
// one alternating; nb == 1 there).
template <bool MULTI = false>
__device__ __forceinline__ void ticket_item(const Args &a, int tk, int lst, int &g, int &r) {
  if (MULTI) {
    g = (((tk / a.R) / a.ndg) * a.nlists + lst) * a.ndg + (tk / a.R) % a.ndg;
    r = tk % a.R;
  } else {
    const int v = (tk / a.bs) * a.nlists + lst;
    g = v / a.nb;
    r = (v - g * a.nb) * a.bs + tk % a.bs;
  }
}

// MULTI kernels: the stream a merged static-dim index belongs to (at most 4 streams, begin[] ascending, unused
// entries = INT_MAX) and the dim's columns there
struct LaneStream { int sd, din, dstat, dout, dvar; };  // dvar: the dim's window-0 column in a global (D,) variance vector
// transposed form (StreamMap::tr_u): merged index d = u * tr_nd + dim of utterance b0 + u; the columns carry the utterance's offset
__device__ __forceinline__ LaneStream lane_stream_tr(const StreamMap &sm, int d) {
  const int u = d / sm.tr_nd, dl = d - u * sm.tr_nd;
  return {sm.sd[0], u * sm.tr_in + sm.in_col[0] + dl, u * sm.tr_stat + sm.stat_col[0] + dl, u * sm.tr_out + sm.out_col[0] + dl,
          sm.in_col[0] + dl};
}
__device__ __forceinline__ LaneStream lane_stream(const StreamMap &sm, int d) {
  const int s_ = (d >= sm.begin[1]) + (d >= sm.begin[2]) + (d >= sm.begin[3]);
  auto pick = [&](const int (&v)[4]) { return s_ == 0 ? v[0] : s_ == 1 ? v[1] : s_ == 2 ? v[2] : v[3]; };
  const int dl = d - pick(sm.begin);
  return {pick(sm.sd), dl + pick(sm.in_col), dl + pick(sm.stat_col), dl + pick(sm.out_col), dl + pick(sm.in_col)};
}

constexpr size_t kLdsStage = (size_t)kStage * kRec * 64 * 8;     // level-3 staging; its head doubles as the level-1 records
constexpr size_t kLdsPark = (size_t)kPark * 64 * 8;
constexpr size_t kLdsFac = (size_t)(kW - 1) * kFac * 64 * 8;
constexpr size_t kLdsU = (size_t)(kW + 1) * 2 * 64 * 8;
constexpr size_t kLdsMisc = 64 + kW * 256;  // control words (+ slack)
constexpr size_t kLdsBytes = kLdsStage + kLdsPark + kLdsFac + kLdsU + kLdsMisc;
static_assert(kW * kRec <= kStage * kRec, "level-1 records must fit the staging area");
static_assert(kLdsBytes <= 160 * 1024 / MLPG_STRIP_WGS, "MLPG_STRIP_WGS workgroups per CU must fit the 160 KB of LDS");

__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}
template <typename T>
__device__ __forceinline__ double tau_of(T v);
template <>
__device__ __forceinline__ double tau_of<float>(float v) {
  return (double)__fdiv_rn(1.0f, v);  // float32 reciprocal, as _mlpg.py:188
}
template <>
__device__ __forceinline__ double tau_of<double>(double v) {
  return fast_rcp(v);
}

// ---- 2x2 blocks, one per lane ---------------------------------------------------------------
struct S2 { double a, b, c; };     // symmetric [a b; b c]
struct M2 { double a, b, c, d; };  // full      [a b; c d]
struct V2 { double x, y; };

__device__ __forceinline__ S2 sym_inv(const S2 &E, bool &bad) {
  const double det = E.a * E.c - E.b * E.b;
  bad |= (E.a <= 0.0) | (det <= 0.0) | !(det == det);
  const double idet = fast_rcp(det);
  return {E.c * idet, -E.b * idet, E.a * idet};
}
__device__ __forceinline__ M2 mul_ms(const M2 &L, const S2 &S) {  // L S
  return {L.a * S.a + L.b * S.b, L.a * S.b + L.b * S.c, L.c * S.a + L.d * S.b, L.c * S.b + L.d * S.c};
}
__device__ __forceinline__ M2 mul_sm(const S2 &S, const M2 &V) {  // S V
  return {S.a * V.a + S.b * V.c, S.a * V.b + S.b * V.d, S.b * V.a + S.c * V.c, S.b * V.b + S.c * V.d};
}
__device__ __forceinline__ M2 mul_smt(const S2 &S, const M2 &V) {  // S V^T
  return {S.a * V.a + S.b * V.b, S.a * V.c + S.b * V.d, S.b * V.a + S.c * V.b, S.b * V.c + S.c * V.d};
}
__device__ __forceinline__ double amax4(const M2 &m) {  // twice this bounds the block's 2-norm
  return __builtin_fmax(__builtin_fmax(__builtin_fabs(m.a), __builtin_fabs(m.b)),
                        __builtin_fmax(__builtin_fabs(m.c), __builtin_fabs(m.d)));
}
__device__ __forceinline__ M2 mul_mm(const M2 &A, const M2 &B) {
  return {A.a * B.a + A.b * B.c, A.a * B.b + A.b * B.d, A.c * B.a + A.d * B.c, A.c * B.b + A.d * B.d};
}
__device__ __forceinline__ S2 mul_mmt_sym(const M2 &A, const M2 &B) {  // A B^T, symmetric by construction
  return {A.a * B.a + A.b * B.b, A.a * B.c + A.b * B.d, A.c * B.c + A.d * B.d};
}
__device__ __forceinline__ S2 mul_mtm_sym(const M2 &A, const M2 &B) {  // A^T B, symmetric by construction
  return {A.a * B.a + A.c * B.c, A.a * B.b + A.c * B.d, A.b * B.b + A.d * B.d};
}
__device__ __forceinline__ V2 mul_mv(const M2 &A, const V2 &v) { return {A.a * v.x + A.b * v.y, A.c * v.x + A.d * v.y}; }
__device__ __forceinline__ V2 mul_mtv(const M2 &A, const V2 &v) { return {A.a * v.x + A.c * v.y, A.b * v.x + A.d * v.y}; }
__device__ __forceinline__ V2 mul_sv(const S2 &S, const V2 &v) { return {S.a * v.x + S.b * v.y, S.b * v.x + S.c * v.y}; }
__device__ __forceinline__ S2 sub(const S2 &A, const S2 &B) { return {A.a - B.a, A.b - B.b, A.c - B.c}; }
__device__ __forceinline__ S2 add(const S2 &A, const S2 &B) { return {A.a + B.a, A.b + B.b, A.c + B.c}; }
__device__ __forceinline__ V2 sub(const V2 &A, const V2 &B) { return {A.x - B.x, A.y - B.y}; }
__device__ __forceinline__ V2 add(const V2 &A, const V2 &B) { return {A.x + B.x, A.y + B.y}; }
__device__ __forceinline__ M2 neg(const M2 &A) { return {-A.a, -A.b, -A.c, -A.d}; }
__device__ __forceinline__ M2 transpose(const M2 &A) { return {A.a, A.c, A.b, A.d}; }

// agent-scope (sc1, write-through / L1-bypassing) 8-byte accesses for the records
__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

// ---- level 1: assembly of one chunk -----------------------------------------------------------
// Frames f0-1 .. f0+M feed the rows f0 .. f0+M-1.  EDGE = false: every one of those frames is a
// live frame of every window (mw <= f0-1, f0+M < T-mw): no masks.  mcol / vcol / gcol point at
// (frame 0, window 0, this lane's dim) of the utterance.
// The 18 frames of a window are taken in six batches of 3 through two register sets: the loads of
// batch k+1 are in flight while batch k is accumulated, so that only the first batch's memory latency
// is exposed (left to itself the compiler issues a window's loads, waits, computes, and only then
// touches the next window).  Small batches because the accumulators already take 128 registers.
constexpr int kNB = kM == 16 ? 6 : 2;  // batches per window (even: every window starts in the same register set)
constexpr int kHB = (kM + 2) / kNB;   // frames per batch
static_assert(kHB * kNB == kM + 2, "batches must tile the 18 frames");

// Loads go through buffer descriptors: a wave-uniform descriptor (the utterance's rows from the dim group's first
// column on), the row/window offset in an SGPR (soffset) and this lane's 32-bit byte offset in ONE VGPR -- no
// per-load 64-bit address arithmetic and no address registers (global_load with 64-bit VGPR addresses costs two VALU
// instructions and a register pair per load, which is what drove this kernel into scratch).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <typename TIN>
__device__ __forceinline__ TIN ld_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
template <>
__device__ __forceinline__ double ld_row<double>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 0);
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <>
__device__ __forceinline__ float ld_row<float>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 0));
}
// the backward's grad_out rows (round 3 measured the `nt` cache policy for them: 2 % slower)
template <typename TIN>
__device__ __forceinline__ TIN ld_row_g(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff);
template <>
__device__ __forceinline__ double ld_row_g<double>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 0);
  return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <>
__device__ __forceinline__ float ld_row_g<float>(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, loff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
  // the base must be wave-uniform PROVABLY (a lane-tainted descriptor is wrapped in a waterfall loop per load)
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// row stores through a buffer descriptor (backward epilogue): scalar row/window offset + one lane offset, no 64-bit
// address arithmetic per store
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const u32x2 w = {(unsigned)u, (unsigned)(u >> 32)};
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, loff, soff, MLPG_STRIP_NT_STORES ? 2 : 0);
}
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned loff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, loff, soff, MLPG_STRIP_NT_STORES ? 2 : 0);
}

// Window coefficients straight from the kernel-argument segment, as scalar loads PLACED by the caller (backward
// kernels, round 5).  The arguments by value are preloaded into scalar registers at kernel entry and live from there to
// their last use; the backward instances need the six products per window in the stream AND the three coefficients per
// window in the epilogue, together with the descriptors and the item loop's state more than the 100 scalar registers
// there are, and the allocator's answer was to park them in lanes of a vector register and fetch them back (v_readlane)
// in front of every use -- some 1 300 extra vector instructions per chunk.  An `asm volatile` load is neither hoisted nor
// shared: the coefficients occupy registers from here to their last use in the same phase, and the scalar cache answers.
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ double mk_f64(unsigned lo, unsigned hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// doubles [first, first + 6) of the argument segment at byte offset `off`
template <unsigned OFF>
__device__ __forceinline__ void karg_f64x6(double (&d)[6]) {
  const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  u32x8 a;
  u32x4 b;
  asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b)
               : "s"(kp), "n"(OFF), "n"(OFF + 32));
  d[0] = mk_f64(a[0], a[1]); d[1] = mk_f64(a[2], a[3]); d[2] = mk_f64(a[4], a[5]); d[3] = mk_f64(a[6], a[7]);
  d[4] = mk_f64(b[0], b[1]); d[5] = mk_f64(b[2], b[3]);
}
typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
template <unsigned OFF>
__device__ __forceinline__ void karg_f64x9(double (&d)[9]) {
  const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  u32x16 a;
  u32x2 b;
  asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b)
               : "s"(kp), "n"(OFF), "n"(OFF + 64));
#pragma unroll
  for (int q = 0; q < 8; ++q) d[q] = mk_f64(a[2 * q], a[2 * q + 1]);
  d[8] = mk_f64(b[0], b[1]);
}
template <unsigned OFF>
__device__ __forceinline__ void karg_f64x3(double (&d)[3]) {
  const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  u32x4 a;
  u32x2 b;
  asm volatile("s_load_dwordx4 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b)
               : "s"(kp), "n"(OFF), "n"(OFF + 16));
  d[0] = mk_f64(a[0], a[1]); d[1] = mk_f64(a[2], a[3]); d[2] = mk_f64(b[0], b[1]);
}
// byte offset of Args::wc in the argument segment of strip_kernel(Problem, WinSet, Args)
constexpr unsigned kKargWs = (unsigned)((sizeof(Problem) + alignof(WinSet) - 1) / alignof(WinSet) * alignof(WinSet));
constexpr unsigned kKargArgs = (unsigned)((kKargWs + sizeof(WinSet) + alignof(Args) - 1) / alignof(Args) * alignof(Args));
constexpr unsigned kKargWc = kKargArgs + (unsigned)__builtin_offsetof(Args, wc);

// EDGE: a dead frame of the window (outside [lo, hi)) is loaded from the nearest live frame instead -- a finite
// variance -- and enters with weight 0 (accumulate()), so that padding values never meet the reciprocal.
template <typename TIN, bool BWD, int VM, bool EDGE, int H>
__device__ __forceinline__ void load_batch(TIN (&rv)[kHB], TIN (&rm)[kHB], __amdgpu_buffer_rsrc_t mrs,
                                           __amdgpu_buffer_rsrc_t vrs, unsigned woff, unsigned loff, unsigned ldi_bytes,
                                           int f0, int lo, int hi) {
#pragma unroll
  for (int q = 0; q < kHB; ++q) {
    int t = f0 + H * kHB + q - 1;
    if (EDGE) t = t < lo ? lo : (t >= hi ? hi - 1 : t);
    const unsigned soff = (unsigned)t * ldi_bytes + woff;
    if (VM == MLPG_HIP_VAR_FRAME) rv[q] = ld_row<TIN>(vrs, soff, loff);
    if (!BWD) rm[q] = ld_row<TIN>(mrs, soff, loff);
  }
}

struct WinCoef {
  double cm, c0, cp, c00, cpp, cmm, cp0, c0m, cpm, tau_glob;
  int w;
};

template <typename TIN, bool BWD, int VM, bool EDGE, int H>
__device__ __forceinline__ void accumulate(const TIN (&rv)[kHB], const TIN (&rm)[kHB], const WinCoef &k, int f0, int lo,
                                           int hi, double (&Pd)[kM], double (&P1)[kM], double (&P2)[kM],
                                           double (&rhs)[kM], double &ca, double &cb, double &cc) {
#pragma unroll
  for (int q = 0; q < kHB; ++q) {
    const int i = H * kHB + q - 1;  // frame f0 + i, compile-time
    const int t = f0 + i;
    double tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(rv[q]) : k.tau_glob;
    if (MLPG_STRIP_ABLATE >= 3) {
      if (i >= 0 && i < kM) { Pd[i] += VM == MLPG_HIP_VAR_FRAME ? (double)rv[q] : 1.0; if (!BWD) rhs[i] += (double)rm[q]; }
      continue;
    }
    if (EDGE) tau *= (t >= lo && t < hi) ? 1.0 : 0.0;  // wave-uniform weight (a scalar select, no branch, no mask)
    double tm = 0.0;
    if (!BWD) tm = tau * (double)rm[q];
    if (i >= 0 && i < kM) {  // row f = t
      Pd[i] += k.c00 * tau;
      P1[i] += k.cp0 * tau;
      if (!BWD) rhs[i] += k.c0 * tm;
    }
    if (i + 1 >= 0 && i + 1 < kM) {  // row f = t+1
      Pd[i + 1] += k.cpp * tau;
      if (!BWD) rhs[i + 1] += k.cp * tm;
    }
    if (i - 1 >= 0 && i - 1 < kM) {  // row f = t-1
      Pd[i - 1] += k.cmm * tau;
      P1[i - 1] += k.c0m * tau;
      P2[i - 1] += k.cpm * tau;
      if (!BWD) rhs[i - 1] += k.cm * tm;
    }
    // coupling of the chunk's first two rows to the previous chunk's separator:
    // ca = P[f0, f0-2], cb = P[f0, f0-1], cc = P[f0+1, f0-1]
    if (i == -1) {
      ca += k.cpm * tau;
      cb += k.cp0 * tau;
    }
    if (i == 0) {
      cb += k.c0m * tau;
      cc += k.cpm * tau;
    }
  }
}

template <typename TIN, int VM>
__device__ __forceinline__ WinCoef win_coef(const double (*wc)[9], int w, const TIN *vglob, int sd) {
  WinCoef k;
  k.cm = wc[w][0]; k.c0 = wc[w][1]; k.cp = wc[w][2];
  k.c00 = wc[w][3]; k.cpp = wc[w][4]; k.cmm = wc[w][5];
  k.cp0 = wc[w][6]; k.c0m = wc[w][7]; k.cpm = wc[w][8];
  k.tau_glob = 1.0;
  if (VM == MLPG_HIP_VAR_GLOBAL) k.tau_glob = tau_of<TIN>(vglob[w * sd]);
  k.w = w;
  return k;
}

template <typename TIN, bool BWD, int VM, bool EDGE>
__device__ __forceinline__ void assemble(__amdgpu_buffer_rsrc_t mrs, __amdgpu_buffer_rsrc_t vrs,
                                         __amdgpu_buffer_rsrc_t grs, const TIN *__restrict__ vglob, unsigned loff,
                                         long ldi, long ldg, int sd, int f0, int T, int Tmax, const WinSet &ws,
                                         const double (*wc)[9], double (&Pd)[kM], double (&P1)[kM], double (&P2)[kM], double (&rhs)[kM],
                                         double &ca, double &cb, double &cc) {
#pragma unroll
  for (int i = 0; i < kM; ++i) Pd[i] = P1[i] = P2[i] = rhs[i] = 0.0;
  ca = cb = cc = 0.0;
  const int mw = ws.mw, nw = ws.nw;
  const unsigned ldi_bytes = (unsigned)ldi * (unsigned)sizeof(TIN), win_bytes = (unsigned)sd * (unsigned)sizeof(TIN);
  // live frames of a window: [0, T) for the static window, [mw, T - mw) for the dynamic ones (none if mw == 0:
  // Python's precisions[-0:] = 0 slice, _mlpg.py:191-193).  The caller guarantees f0 < T.
  auto live_lo = [&](int w) { return w ? mw : 0; };
  auto live_hi = [&](int w) { return w ? (mw != 0 && T - mw > mw ? T - mw : mw) : T; };  // hi == lo: nothing live
  TIN rvA[kHB], rmA[kHB], rvB[kHB], rmB[kHB];
  // a window without live frames (hi == lo) still loads (clamped to frame lo, which exists: lo < T) and weighs 0
  auto clo = [&](int w) { const int l_ = live_lo(w); return l_ < T ? l_ : T - 1; };
  auto chi = [&](int w) { const int h_ = live_hi(w), l_ = clo(w); return h_ > l_ ? h_ : l_ + 1; };
  load_batch<TIN, BWD, VM, EDGE, 0>(rvA, rmA, mrs, vrs, 0u, loff, ldi_bytes, f0, clo(0), chi(0));
  if (BWD) {
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      int t = f0 + i;
      if (EDGE) t = t >= T ? T - 1 : t;
      rhs[i] = (double)ld_row_g<TIN>(grs, (unsigned)t * (unsigned)ldg * (unsigned)sizeof(TIN), loff);  // rows >= T are reset below
    }
  }
  for (int w = 0; w < nw; ++w) {
    const WinCoef k = win_coef<TIN, VM>(wc, w, vglob, sd);
    const unsigned woff = (unsigned)w * win_bytes;
    const int lo = live_lo(w), hi = live_hi(w), cl = clo(w), ch = chi(w);
    // the six batches alternate between the two register sets; the window ends with the next window's batch 0 in
    // flight in set A.  sched_barrier pins the issue order: next batch's loads, then this batch's arithmetic.
#define STRIP_STEP(H, RVX, RMX, RVY, RMY)                                                                      \
    if ((H) + 1 < kNB) {                                                                                          \
    load_batch<TIN, BWD, VM, EDGE, ((H) + 1 < kNB ? (H) + 1 : 0)>(RVY, RMY, mrs, vrs, woff, loff, ldi_bytes, f0, cl, ch);             \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    accumulate<TIN, BWD, VM, EDGE, (H)>(RVX, RMX, k, f0, lo, hi, Pd, P1, P2, rhs, ca, cb, cc);                  \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    STRIP_STEP(0, rvA, rmA, rvB, rmB)
    STRIP_STEP(1, rvB, rmB, rvA, rmA)
    STRIP_STEP(2, rvA, rmA, rvB, rmB)
    STRIP_STEP(3, rvB, rmB, rvA, rmA)
    STRIP_STEP(4, rvA, rmA, rvB, rmB)
#undef STRIP_STEP
    if (w + 1 < nw) load_batch<TIN, BWD, VM, EDGE, 0>(rvA, rmA, mrs, vrs, woff + win_bytes, loff, ldi_bytes, f0, clo(w + 1), chi(w + 1));
    __builtin_amdgcn_sched_barrier(0);
    accumulate<TIN, BWD, VM, EDGE, kNB - 1>(rvB, rmB, k, f0, lo, hi, Pd, P1, P2, rhs, ca, cb, cc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EDGE) {
    // matrix edges: rows >= T are identity rows, entries that would leave the T x T matrix vanish
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      const int f = f0 + i;
      if (f >= T) {
        Pd[i] = 1.0;
        P1[i] = P2[i] = rhs[i] = 0.0;
      } else {
        if (f + 1 >= T) P1[i] = 0.0;
        if (f + 2 >= T) P2[i] = 0.0;
      }
    }
    if (f0 == 0 || f0 >= T) ca = cb = cc = 0.0;
    else if (f0 + 1 >= T) cc = 0.0;
  }
}

// ---- level 1: interior elimination (as solve_chunk of mlpg_wave_impl.h, without the lane shuffles) ----
// On return Pd/P1/P2/rhs[0..kN) hold 1/d, l1, l2, g; rec[] the chunk's Schur data.
__device__ __forceinline__ bool eliminate(double (&Pd)[kM], double (&P1)[kM], double (&P2)[kM], double (&rhs)[kM],
                                          double ca, double cb, double cc, double (&rec)[kRec]) {
  bool bad = false;
  double t00 = 0.0, t01 = 0.0, t11 = 0.0, h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
#pragma unroll
  for (int i = 0; i < kN; ++i) {
    const double dd = Pd[i];
    bad |= !(dd > 0.0);
    const double dinv = fast_rcp(dd);
    const double e1 = P1[i], e2 = P2[i];
    const double l1 = e1 * dinv, l2 = e2 * dinv;
    Pd[i + 1] -= l1 * e1;
    P1[i + 1] -= l2 * e1;
    Pd[i + 2] -= l2 * e2;
    const double gi = rhs[i] - l1p * g1 - l2pp * g2;
    const double ba = (i == 0) ? ca : 0.0;
    const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
    const double va = ba - l1p * va1 - l2pp * va2;
    const double vb = bb - l1p * vb1 - l2pp * vb2;
    const double wa = va * dinv, wb = vb * dinv;
    t00 += wa * va;
    t01 += wa * vb;
    t11 += wb * vb;
    h0 += wa * gi;
    h1 += wb * gi;
    Pd[i] = dinv;
    P1[i] = l1;
    P2[i] = l2;
    rhs[i] = gi;
    g2 = g1; g1 = gi;
    va2 = va1; va1 = va;
    vb2 = vb1; vb1 = vb;
    l2pp = l2p; l2p = l2; l1p = l1;
    // the recurrence is sequential; letting the scheduler interleave the frames only inflates the live set
    __builtin_amdgcn_sched_barrier(0);
  }
  rec[rT00] = t00; rec[rT01] = t01; rec[rT11] = t11; rec[rH0] = h0; rec[rH1] = h1;
  rec[rD11] = Pd[kN]; rec[rD12] = P1[kN]; rec[rD22] = Pd[kN + 1];
  rec[rF1] = rhs[kN] - (l1p * g1 + l2pp * g2);
  rec[rF2] = rhs[kN + 1] - l2p * g1;
  rec[rL11] = -(l1p * va1 + l2pp * va2);
  rec[rL12] = -(l1p * vb1 + l2pp * vb2);
  rec[rL21] = -(l2p * va1);
  rec[rL22] = -(l2p * vb1);
  return bad;
}

// ---- level 1, streamed: assembly and interior elimination of one chunk in frame order (NW windows, known at
// compile time) ----
// The 18 frames f0-1 .. f0+16 are taken one at a time; a frame is NW rows of `var` and of `mean` (one load each per
// lane) and feeds matrix rows t-1, t, t+1.  Row i is therefore final once frame i+1 has been accumulated, and is
// eliminated right then: the recurrence of `eliminate` runs under the loads instead of after them.  A ring of kRing
// frames is in flight from the first load to the last: as soon as a frame has been accumulated its registers are
// refilled with the frame kRing further on.  Unlike the window-major order of `assemble`, rows that have not been
// reached yet hold nothing, so the ring and the accumulators never peak together: this is what lets the loads
// stream without spilling.  Arithmetic per entry: the same sums in a different order (frame-major).
template <typename TIN>
struct RingDepth { static constexpr int value = MLPG_STRIP_RING_F64; };   // frames of loads in flight per wavefront (float64: 36 loads, 18 KB)
template <>
struct RingDepth<float> { static constexpr int value = MLPG_STRIP_RING_F32; };  // float32 values take half the registers
// LT (the transposed form with a lengths vector): T is the frame count of the lane group's LONGEST utterance -- what the clamped loads
// and everything wave-uniform go by -- and Tu this lane's own: its dead frames enter with precision 0 and mean 0 by per-lane
// SELECTS (their values are padding: anything), its rows >= Tu become identity rows.
template <typename TIN, bool BWD, int VM, bool EDGE, int NW, bool MULTI = false, bool KEEP = false, bool LT = false>
__device__ __forceinline__ bool assemble_eliminate(__amdgpu_buffer_rsrc_t mrs, __amdgpu_buffer_rsrc_t vrs,
                                                   __amdgpu_buffer_rsrc_t grs, const TIN *__restrict__ vglob,
                                                   unsigned loff, long ldi, long ldg, int sd, int f0, int T, int mw,
                                                   const double (*wc)[9], const double one, double (&Pd)[kM], double (&P1)[kM],
                                                   double (&P2)[kM], double (&rhs)[kM], double &ca, double &cb,
                                                   double &cc, double (&rec)[kRec], float (&tk)[kM + 1][NW], const int Tu = 0) {
  // KEEP (backward, float32 inputs): tk[i + 1][w] = the precision of frame f0 + i in window w as the assembly used it
  // (dead frames 0), i = -1 .. kM-1: what the epilogue multiplies the gradient rows with
  // No zero-fill: every accumulator is ASSIGNED by the first contribution that reaches it (window 0 of the frame
  // noted below), so that a row costs no register before its first frame arrives.
  const unsigned ldi_bytes = (unsigned)ldi * (unsigned)sizeof(TIN), win_bytes = (unsigned)sd * (unsigned)sizeof(TIN);
  // live frames of a window: [0, T) for the static window, [mw, T - mw) for the dynamic ones (none if mw == 0)
  int lo[NW], hi[NW], cl[NW], ch[NW];
  int hi_l[NW];  // LT: this lane's own end of the window's live frames
  WinCoef k[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    lo[w] = w ? mw : 0;
    hi[w] = w ? (mw != 0 && T - mw > mw ? T - mw : mw) : T;
    hi_l[w] = w ? (mw != 0 && Tu - mw > mw ? Tu - mw : mw) : Tu;
    cl[w] = lo[w] < T ? lo[w] : T - 1;            // a window without live frames still loads (frame cl) and weighs 0
    ch[w] = hi[w] > cl[w] ? hi[w] : cl[w] + 1;
    k[w] = win_coef<TIN, VM>(wc, w, vglob, sd);
    // unit variances: the precision 1.0 as an opaque per-lane run-time value.  As a literal (or any wave-uniform value)
    // the whole matrix becomes uniform arithmetic that the compiler hoists above the stream and spills (544-880 B/lane).
    if (VM == MLPG_HIP_VAR_UNIT) {
      double t1 = one;
      asm volatile("" : "+v"(t1));  // a per-lane value as far as the compiler can tell
      k[w].tau_glob = t1;
    }
  }
  constexpr int kRing = RingDepth<TIN>::value;
  TIN rv[kRing][NW], rm[kRing][NW];
  auto load_frame = [&](TIN (&v)[NW], TIN (&m)[NW], const int i) __attribute__((always_inline)) {
    // BWD: the frame's row offset as an opaque scalar, so that the 18 multiples i * ldi_bytes (and their EDGE variants) are
    // not hoisted out of the persistent item loop as invariants that occupy scalar registers through the whole kernel.
    // (Part of round 5's hunt for the backward instances' scalar spills; what removed the v_readlane reloads in the stream
    // were the placed coefficient loads -- karg_f64x6 -- this alone did not.)
    unsigned frame_off = (unsigned)(f0 + i) * ldi_bytes;
    if (BWD && !EDGE) asm volatile("" : "+s"(frame_off));
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int t = f0 + i;
      if (EDGE) {
        t = t < cl[w] ? cl[w] : (t >= ch[w] ? ch[w] - 1 : t);
        frame_off = (unsigned)t * ldi_bytes;
        if (BWD) asm volatile("" : "+s"(frame_off));
      }
      // MULTI: `sd` is this LANE's window pitch (the static dim of its stream), so the window offset is part of the
      // lane offset; otherwise it is wave-uniform and rides in the scalar offset
      const unsigned soff = MULTI ? frame_off : frame_off + (unsigned)w * win_bytes;
      const unsigned voff = MULTI ? loff + (unsigned)w * win_bytes : loff;
      if (VM == MLPG_HIP_VAR_FRAME) v[w] = ld_row<TIN>(vrs, soff, voff);
      if (!BWD) m[w] = ld_row<TIN>(mrs, soff, voff);
    }
  };
  auto accumulate_frame = [&](const TIN (&v)[NW], const TIN (&m)[NW], const int i) __attribute__((always_inline)) {
    const int t = f0 + i;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      double tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(v[w]) : k[w].tau_glob;
      double mval = BWD ? 0.0 : (double)m[w];
      if (EDGE && LT) {
        const bool lv = t >= lo[w] && t < hi_l[w];  // per lane
        tau = lv ? tau : 0.0;
        mval = lv ? mval : 0.0;
      } else if (EDGE) {
        tau *= (t >= lo[w] && t < hi[w]) ? 1.0 : 0.0;  // wave-uniform weight
      }
      if (KEEP && i < kM) tk[i + 1][w] = (float)tau;  // float32 inputs: exact (a float32 reciprocal, or 0); float64 inputs: rounded
      double tm = 0.0;
      if (!BWD) tm = tau * mval;
      const bool first = w == 0;  // first contribution to: Pd, rhs of row t+1; P1 of row t; P2 of row t-1
      if (i >= 0 && i < kM) {  // row t
        Pd[i] += k[w].c00 * tau;
        P1[i] = first ? k[w].cp0 * tau : P1[i] + k[w].cp0 * tau;
        if (!BWD) rhs[i] += k[w].c0 * tm;
      }
      if (i + 1 >= 0 && i + 1 < kM) {  // row t+1
        Pd[i + 1] = first ? k[w].cpp * tau : Pd[i + 1] + k[w].cpp * tau;
        if (!BWD) rhs[i + 1] = first ? k[w].cp * tm : rhs[i + 1] + k[w].cp * tm;
      }
      if (i - 1 >= 0 && i - 1 < kM) {  // row t-1
        Pd[i - 1] += k[w].cmm * tau;
        P1[i - 1] += k[w].c0m * tau;
        P2[i - 1] = first ? k[w].cpm * tau : P2[i - 1] + k[w].cpm * tau;
        if (!BWD) rhs[i - 1] += k[w].cm * tm;
      }
      // coupling of the chunk's first two rows to the previous chunk's separator:
      // ca = P[f0, f0-2], cb = P[f0, f0-1], cc = P[f0+1, f0-1]
      if (i == -1) {
        ca = first ? k[w].cpm * tau : ca + k[w].cpm * tau;
        cb = first ? k[w].cp0 * tau : cb + k[w].cp0 * tau;
      }
      if (i == 0) {
        cb += k[w].c0m * tau;
        cc = first ? k[w].cpm * tau : cc + k[w].cpm * tau;
      }
    }
  };
  // matrix edges (EDGE): rows >= T are identity rows, entries that would leave the T x T matrix vanish.  Wave-uniform
  // 0/1 factors instead of branches (x * 1 and x * 1 + 0 are exact; dead frames enter with weight 0, so nothing
  // that is multiplied by 0 here can be Inf or NaN unless the input is): straight-line code for the allocator.
  auto fix_row = [&](const int i) __attribute__((always_inline)) {
    const int f = f0 + i;
    if (LT) {
      Pd[i] = f < Tu ? Pd[i] : 1.0;
      P1[i] = f + 1 < Tu ? P1[i] : 0.0;
      P2[i] = f + 2 < Tu ? P2[i] : 0.0;
      rhs[i] = f < Tu ? rhs[i] : 0.0;
      return;
    }
    const double live = f < T ? 1.0 : 0.0, live1 = f + 1 < T ? 1.0 : 0.0, live2 = f + 2 < T ? 1.0 : 0.0;
    Pd[i] = Pd[i] * live + (1.0 - live);
    P1[i] *= live1;
    P2[i] *= live2;
    rhs[i] *= live;
  };
  // elimination state (see `eliminate`)
  bool bad = false;
  double t00 = 0.0, t01 = 0.0, t11 = 0.0, h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
  auto elim_row = [&](const int i) __attribute__((always_inline)) {
    if (EDGE) {
      fix_row(i);
      if (i == 0 && LT) {
        const bool keep = !(f0 == 0 || f0 >= Tu), keepc = keep && !(f0 + 1 >= Tu);
        ca = keep ? ca : 0.0;
        cb = keep ? cb : 0.0;
        cc = keepc ? cc : 0.0;
      } else if (i == 0) {
        const double keep = (f0 == 0 || f0 >= T) ? 0.0 : 1.0, keepc = f0 + 1 >= T ? 0.0 : 1.0;
        ca *= keep;
        cb *= keep;
        cc *= keep * keepc;
      }
    }
    const double dd = Pd[i];
    bad |= !(dd > 0.0);
    const double dinv = fast_rcp(dd);
    const double e1 = P1[i], e2 = P2[i];
    const double l1 = e1 * dinv, l2 = e2 * dinv;
    Pd[i + 1] -= l1 * e1;
    P1[i + 1] -= l2 * e1;
    Pd[i + 2] -= l2 * e2;
    const double gi = rhs[i] - l1p * g1 - l2pp * g2;
    const double ba = (i == 0) ? ca : 0.0;
    const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
    const double va = ba - l1p * va1 - l2pp * va2;
    const double vb = bb - l1p * vb1 - l2pp * vb2;
    const double wa = va * dinv, wb = vb * dinv;
    t00 += wa * va;
    t01 += wa * vb;
    t11 += wb * vb;
    h0 += wa * gi;
    h1 += wb * gi;
    Pd[i] = dinv;
    P1[i] = l1;
    P2[i] = l2;
    rhs[i] = gi;
    g2 = g1; g1 = gi;
    va2 = va1; va1 = va;
    vb2 = vb1; vb1 = vb;
    l2pp = l2p; l2p = l2; l1p = l1;
  };

  // prologue: the first kRing frames (f0-1 ..) in flight
#pragma unroll
  for (int sl = 0; sl < kRing; ++sl) load_frame(rv[sl], rm[sl], sl - 1);
  static_assert(kRing >= 2 && kRing <= kM + 2, "ring depth");
  if (BWD) {
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      int t = f0 + i;
      if (EDGE) t = t >= T ? T - 1 : t;
      rhs[i] = (double)ld_row_g<TIN>(grs, (unsigned)t * (unsigned)ldg * (unsigned)sizeof(TIN), loff);  // rows >= T are reset by fix_row
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // step S handles frame I = S - 1 in ring slot S % 6: accumulate, refill with frame I + 6, eliminate row I - 1
  // (the three barriers pin accumulation / refill / elimination in this order; relaxing any of them -- round 3 -- changed nothing
  // with two wavefronts per SIMD)
#define STRIP_SB(bit) __builtin_amdgcn_sched_barrier(0)
#define STRIP_STEP(S)                                                                   \
  if ((S) < kM + 2) {                                                                   \
  accumulate_frame(rv[(S) % kRing], rm[(S) % kRing], (S)-1);                            \
  STRIP_SB(1);                                                                          \
  if ((S) + kRing < kM + 2) { load_frame(rv[(S) % kRing], rm[(S) % kRing], (S)-1 + kRing); } \
  STRIP_SB(2);                                                                          \
  if ((S)-2 >= 0 && (S)-2 < kN) { elim_row((S)-2); }                                    \
  STRIP_SB(4);                                                                          \
  }
  STRIP_STEP(0) STRIP_STEP(1) STRIP_STEP(2) STRIP_STEP(3) STRIP_STEP(4) STRIP_STEP(5)
  STRIP_STEP(6) STRIP_STEP(7) STRIP_STEP(8) STRIP_STEP(9) STRIP_STEP(10) STRIP_STEP(11)
  STRIP_STEP(12) STRIP_STEP(13) STRIP_STEP(14) STRIP_STEP(15) STRIP_STEP(16) STRIP_STEP(17)
#undef STRIP_STEP
#undef STRIP_SB
  if (EDGE) {
    fix_row(kN);
    fix_row(kN + 1);
  }
  rec[rT00] = t00; rec[rT01] = t01; rec[rT11] = t11; rec[rH0] = h0; rec[rH1] = h1;
  rec[rD11] = Pd[kN]; rec[rD12] = P1[kN]; rec[rD22] = Pd[kN + 1];
  rec[rF1] = rhs[kN] - (l1p * g1 + l2pp * g2);
  rec[rF2] = rhs[kN + 1] - l2p * g1;
  rec[rL11] = -(l1p * va1 + l2pp * va2);
  rec[rL12] = -(l1p * vb1 + l2pp * vb2);
  rec[rL21] = -(l2p * va1);
  rec[rL22] = -(l2p * vb1);
  return bad;
}

// ---- level 1: back-substitution; on return x[0..kM) is the chunk's solution -------------------
__device__ __forceinline__ void backsub(const double (&Pd)[kM], const double (&P1)[kM], const double (&P2)[kM],
                                        double (&rhs)[kM], double ca, double cb, double cc, V2 ul, V2 u) {
  {
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
    for (int i = 0; i < kN; ++i) {
      const double ba = (i == 0) ? ca : 0.0;
      const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
      const double va = ba - q1 * a1 - q3 * a2;
      const double vb = bb - q1 * b1 - q3 * b2;
      rhs[i] -= va * ul.x + vb * ul.y;
      a2 = a1; a1 = va;
      b2 = b1; b1 = vb;
      q3 = q2; q2 = P2[i]; q1 = P1[i];
    }
  }
  double x1 = u.x, x2 = u.y;
#pragma unroll
  for (int i = kN - 1; i >= 0; --i) {
    const double xi = rhs[i] * Pd[i] - P1[i] * x1 - P2[i] * x2;
    rhs[i] = xi;
    x2 = x1;
    x1 = xi;
  }
  rhs[kN] = u.x;
  rhs[kN + 1] = u.y;
}

// ---- the kernel ---------------------------------------------------------------------------------
// NW3: the launch has exactly three windows (the usual static / delta / delta-delta set): level 1 is the streamed
// assemble_eliminate and the backward epilogue the frame-major one, and nothing of the window-major forms for other window
// counts is compiled into the kernel (round 5: those forms alone cost the backward instances ~280 spilled scalar
// registers and 20 k instructions of code); NW3 = false serves one, two or more than three windows.
// TR (MULTI kernels): the lanes of a group run over several utterances of one narrow stream (StreamMap::tr_u).
template <typename TIN, typename TOUT, bool BWD, int VM, bool MULTI = false, bool NW3 = true, bool TR = false>
__global__ __launch_bounds__(kW * 64, MLPG_STRIP_WGS) void strip_kernel(Problem p, WinSet ws, Args a) {
  static_assert(!TR || MULTI, "the transposed form is a MULTI kernel");
  extern __shared__ __align__(16) unsigned char smem[];
  double *lds_rec = (double *)smem;                                  // [kW][kRec][64]   (level 1 -> 2)
  double *lds_stage = (double *)smem;                                // [kStage][kRec][64] (level 3), same bytes
  double *lds_park = (double *)(smem + kLdsStage);                   // [kPark][64]: g and l2 of wavefront 0's chunk during levels 2-3
  double *lds_fac = (double *)(smem + kLdsStage + kLdsPark);         // [kW-1][kFac][64]
  double *lds_u = (double *)(smem + kLdsStage + kLdsPark + kLdsFac); // [kW+1][2][64]: slot j+1 = separator j, slot 0 = previous strip's
  int *lds_misc = (int *)(smem + kLdsStage + kLdsPark + kLdsFac + kLdsU);  // [0] item, [1] poll result

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 16) lds_misc[tid] = 0;
  __syncthreads();

#ifdef MLPG_STRIP_TIMING
  // claim, assemble, eliminate, barrier, level 2, publish, poll, barrier, level-3 staging, level-3 sweep, level-2 backsub, barrier, backsub, store
  long long tq[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t_prev = (long long)__builtin_readcyclecounter();
#endif
  // ---- work items: (system group, strip), handed out by atomic tickets in that order, one list per XCD.
  // A workgroup draws from the list of the XCD it runs on (HW_REG_XCC_ID) until that list is empty, then helps
  // with the other lists: which workgroup runs which item never matters for the result (all inter-workgroup
  // traffic is agent scope), only for speed.
  const int xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;  // hwreg(HW_REG_XCC_ID, 0, 4)
  const int R = a.R;
  auto body = [&](const int g, const int r) __attribute__((always_inline)) {
  // (TR: group g = the block of utterances b .. b + tr_u - 1, one dim group)
  const int b = TR ? g * a.sm.tr_u : g / a.ndg, dg = TR ? 0 : g - b * a.ndg;
  const int Tmax = p.Tmax;
  const long ldi = p.ld_in, ldg = p.ld_gout, ldo = p.ld_out;
  int T = (p.lengths && !TR) ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  int Ract = (T + kW * kM - 1) / (kW * kM);  // strips of this utterance that hold live frames
  bool xwg = MLPG_STRIP_ABLATE ? false : Ract > 1;  // the utterance spans several strips: level 3 runs
  const int d0 = dg * a.dgw;
  const int sd_all = MULTI ? a.sm.total : p.sd;  // MULTI: the lanes run over the static dims of all streams
  const int nd = TR ? (a.sm.tr_B - b < a.sm.tr_u ? a.sm.tr_B - b : a.sm.tr_u) * a.sm.tr_nd
                    : (sd_all - d0 < a.dgw ? sd_all - d0 : a.dgw);
  const bool lane_ok = lane < nd;
  int d = d0 + (lane_ok ? lane : nd - 1);  // idle lanes shadow the group's last dim (never stored)
  // TR with a lengths vector: Tu = this lane's utterance's frame count; T = the longest of the lane group (what every wave-uniform
  // rule goes by: strips, clamped loads, stores), Tmin the shortest (chunks that are interior for EVERY lane need no masks)
  int Tu = T, Tmin = T;
  if (TR && p.lengths) {
    Tu = p.lengths[b + d / a.sm.tr_nd];
    Tu = Tu < 0 ? 0 : (Tu > Tmax ? Tmax : Tu);
    int tmx = Tu, tmn = Tu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const int o1 = __shfl_xor(tmx, off, 64), o2 = __shfl_xor(tmn, off, 64);
      tmx = o1 > tmx ? o1 : tmx;
      tmn = o2 < tmn ? o2 : tmn;
    }
    T = __builtin_amdgcn_readfirstlane(tmx);
    Tmin = __builtin_amdgcn_readfirstlane(tmn);
    Ract = (T + kW * kM - 1) / (kW * kM);
    xwg = MLPG_STRIP_ABLATE ? false : Ract > 1;
  }
  // sd: the pitch between a dim's windows in a row; d: its output (and status) column; din: its window-0 input column
  int sd = p.sd, din = d, dstat = d, dvar = d;
  if (MULTI) {
    const LaneStream ls = TR ? lane_stream_tr(a.sm, d) : lane_stream(a.sm, d);
    sd = ls.sd;
    din = ls.din;
    dstat = ls.dstat;
    dvar = ls.dvar;
    d = ls.dout;
  }
  const int f0 = (r * kW + wv) * kM;
  const int nw = NW3 ? 3 : ws.nw, mw = ws.mw;

  TOUT *out_b = (TOUT *)p.out + (size_t)b * Tmax * ldo;

  if (r >= Ract) {
    // nothing but padding frames here: zero-fill this chunk's rows
    if (lane_ok) {
      for (int i = 0; i < kM; ++i) {
        const int t = f0 + i;
        if (t >= Tmax) break;
        if (!BWD) {
          out_b[(size_t)t * ldo + d] = (TOUT)0;
        } else {
          for (int w = 0; w < nw; ++w) out_b[(size_t)t * ldo + w * sd + d] = (TOUT)0;
        }
      }
    }
#if !defined(MLPG_STRIP_TIMING) && !defined(MLPG_STRIP_TRACE)
    if (r == 0 && wv == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + dstat] = 0;  // T == 0
#endif
    return;
  }

  // wave-uniform bases (utterance b, frame 0, first dim of the group) + this lane's byte offset
  // (MULTI: the base is column 0 of the parent array and the lane offset the dim's absolute input column)
  const int dbase = MULTI ? 0 : d0;
  const unsigned loff = (unsigned)(din - dbase) * (unsigned)sizeof(TIN);
  const __amdgpu_buffer_rsrc_t mrs = make_rsrc(BWD ? (const TIN *)p.out : (const TIN *)p.mean + (size_t)b * Tmax * ldi + dbase);
  const __amdgpu_buffer_rsrc_t vrs =
      make_rsrc(VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)b * Tmax * ldi + dbase : (const TIN *)p.out);
  const TIN *vglob = VM == MLPG_HIP_VAR_GLOBAL ? (const TIN *)p.var + dvar : nullptr;
  const __amdgpu_buffer_rsrc_t grs = make_rsrc(BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * ldg + d0 : (const TIN *)p.out);
  const TIN *vcol = VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)b * Tmax * ldi + d : nullptr;  // backward epilogue
  (void)vcol;
#ifdef MLPG_STRIP_TIMING
  const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz constant clock, comparable across CUs
#endif
  STRIP_TICK(0);
#ifdef MLPG_STRIP_TRACE
  const long long tr0 = (long long)__builtin_amdgcn_s_memrealtime();
  long long tr1 = tr0, tr2 = tr0, tra = tr0;
#endif
  // ---- level 1 ----
  double Pd[kM], P1[kM], P2[kM], rhs[kM], ca, cb, cc;
  double rec[kRec];
  bool bad = false;
  // Backward epilogue inputs (frame-major form, three windows): the precisions of the frames f0-1 .. f0+kM-1.
  //   float32 inputs, or a float32 gradient (kKeepTau): kept from the assembly in tk[][] -- float32 values, 51 registers;
  //   float64 inputs with a float64 gradient: the variance rows are read a second time; those of the first kEarly frames are requested right
  //   after level 1 (early_issue(): the ring's registers are free then), so that they travel while the strip waits
  //   for its neighbours, the rest in front of the epilogue's first store.
  // (float64 inputs with a float32 gradient -- what the reference's mlpg_grad returns, _mlpg.py:248 -- keep the precisions rounded
  // to float32 too: the product is rounded to float32 anyway, so this costs at most one more rounding of 2^-24)
  constexpr bool kKeepTau = NW3 && BWD && VM == MLPG_HIP_VAR_FRAME && (sizeof(TIN) == 4 || sizeof(TOUT) == 4) && MLPG_STRIP_BWD_KEEP && !MULTI;
  constexpr int kEpi = kM + 1;
  constexpr int kEarly = (NW3 && BWD && VM == MLPG_HIP_VAR_FRAME && sizeof(TIN) == 8 && !kKeepTau && MLPG_STRIP_BWD_FRAME_MAJOR) ? MLPG_STRIP_BWD_EARLY : 0;
  // the tail is instantiated per role only where the roles differ in what they keep (otherwise once, behind the roles:
  // two copies of the epilogue cost scalar registers that the streamed level 1 then spills)
  constexpr bool kSplitTail = kKeepTau && !MLPG_STRIP_BWD_KEEP0;
  constexpr int kEarly0 = kSplitTail ? MLPG_STRIP_BWD_EARLY0 : kEarly;  // wavefront 0 (runs levels 2-3 meanwhile)
  static_assert(kEarly >= 0 && kEarly <= kEpi && kEarly0 >= 0 && kEarly0 <= kEpi, "MLPG_STRIP_BWD_EARLY");
  float tk[kEpi][3];
  TIN tv[kEpi][3];
  auto ldf = [&](TIN (&v)[3], const int i) __attribute__((always_inline)) {
    if (VM != MLPG_HIP_VAR_FRAME) return;
    int t = f0 + i;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);  // in bounds (T >= 1 here); a row that is not live is not used
    const unsigned ldi_b = (unsigned)ldi * (unsigned)sizeof(TIN), win_b = (unsigned)sd * (unsigned)sizeof(TIN);
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      v[w] = ld_row<TIN>(vrs, (unsigned)t * ldi_b + (unsigned)w * win_b, loff);  // (only the NW3 kernel comes here)
    }
  };
  auto early_issue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < kEarly; ++sl) ldf(tv[sl], sl - 1);
  };
  auto early_issue0 = [&]() __attribute__((always_inline)) {  // wavefront 0
#pragma unroll
    for (int sl = 0; sl < kEarly0; ++sl) ldf(tv[sl], sl - 1);
  };
  if (f0 < T) {
    const bool interior = mw != 0 && f0 - 1 >= mw && f0 + kM < (TR ? Tmin : T) - mw;
    if (NW3 && MLPG_STRIP_ABLATE < 2) {
      // the usual three windows: assembly and elimination streamed in frame order
      double wcl[3][9];
      const double (*wcs)[9] = a.wc;
      if (BWD && MLPG_STRIP_BWD_KARG) {
        // the six coefficient products per window the backward stream uses (entries 3 .. 8), fetched here
        double c6[3][6];
        karg_f64x6<kKargWc + 0 * 72 + 24>(c6[0]);
        karg_f64x6<kKargWc + 1 * 72 + 24>(c6[1]);
        karg_f64x6<kKargWc + 2 * 72 + 24>(c6[2]);
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          wcl[w][0] = wcl[w][1] = wcl[w][2] = 0.0;
#pragma unroll
          for (int q = 0; q < 6; ++q) wcl[w][3 + q] = c6[w][q];
        }
        wcs = wcl;
      }
      if (!BWD && MLPG_STRIP_FWD_KARG && !MULTI) {
        karg_f64x9<kKargWc + 0 * 72>(wcl[0]);
        karg_f64x9<kKargWc + 1 * 72>(wcl[1]);
        karg_f64x9<kKargWc + 2 * 72>(wcl[2]);
        wcs = wcl;
      }
      if (interior) bad = assemble_eliminate<TIN, BWD, VM, false, 3, MULTI, kKeepTau>(mrs, vrs, grs, vglob, loff, ldi, ldg, sd, f0, T, mw, wcs, a.one, Pd, P1, P2, rhs, ca, cb, cc, rec, tk);
      else bad = assemble_eliminate<TIN, BWD, VM, true, 3, MULTI, kKeepTau, TR>(mrs, vrs, grs, vglob, loff, ldi, ldg, sd, f0, T, mw, wcs, a.one, Pd, P1, P2, rhs, ca, cb, cc, rec, tk, Tu);
      STRIP_TICK(1);
#ifdef MLPG_STRIP_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tra = (long long)__builtin_amdgcn_s_memrealtime();   // this wavefront's loads have all landed
#endif
    } else {
    if (interior) assemble<TIN, BWD, VM, false>(mrs, vrs, grs, vglob, loff, ldi, ldg, sd, f0, T, Tmax, ws, a.wc, Pd, P1, P2, rhs, ca, cb, cc);
    else assemble<TIN, BWD, VM, true>(mrs, vrs, grs, vglob, loff, ldi, ldg, sd, f0, T, Tmax, ws, a.wc, Pd, P1, P2, rhs, ca, cb, cc);
    STRIP_TICK(1);
#ifdef MLPG_STRIP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tra = (long long)__builtin_amdgcn_s_memrealtime();   // this wavefront's loads have all landed
#endif
    if (MLPG_STRIP_ABLATE < 2) bad = eliminate(Pd, P1, P2, rhs, ca, cb, cc, rec);
    else {
#pragma unroll
      for (int k = 0; k < kRec; ++k) rec[k] = Pd[k] + rhs[k];
      rec[rD11] = rec[rD22] = 1.0; rec[rD12] = 0.0;
    }
    }
  } else {
    // a chunk of identity rows behind the utterance's end (keeps the strip's separator chain regular)
#pragma unroll
    for (int i = 0; i < kM; ++i) { Pd[i] = 1.0; P1[i] = P2[i] = rhs[i] = 0.0; }
    ca = cb = cc = 0.0;
#pragma unroll
    for (int k = 0; k < kRec; ++k) rec[k] = 0.0;
    rec[rD11] = rec[rD22] = 1.0;
    if (kKeepTau) {
#pragma unroll
      for (int i = 0; i < kEpi; ++i) tk[i][0] = tk[i][1] = tk[i][2] = 0.0f;
    }
  }
  if (bad) rec[rD11] = __builtin_nan("");  // poisons every later level: the system is reported, not solved
#pragma unroll
  for (int k = 0; k < kRec; ++k) lds_rec[(wv * kRec + k) * 64 + lane] = rec[k];
  if (kEarly > 0 && wv != 0) early_issue();  // (wavefront 0: behind level 3 -- its chain has no registers to spare)
  STRIP_TICK(2);
  __syncthreads();
  STRIP_TICK(3);

  // ---- levels 2 and 3.  Wavefront 0 runs the sequential parts (the strip's separators, then the sweep over
  // the utterance's records); wavefronts 1..3 stage the records for it.  The two roles are separate code paths
  // that meet at the same barriers.  Wavefront 0 parks g and l2 of its own chunk in LDS for the duration: its
  // chains then run out of registers, not out of scratch.
  int timed_out = 0;
  // ---- the item's tail: level-1 back-substitution, verdict marks, output.  A lambda, called at the end of BOTH role
  // branches below (the roles meet at the same barriers), so that what the epilogue keeps in registers across levels 2-3
  // is a per-role matter: wavefront 0 runs the chain and has no register to spare, wavefronts 1 .. kW-1 only wait.
  //   USE_TK: the precisions kept from the assembly (float32 inputs);  NEARLY: variance rows already requested (float64)
  auto tail = [&](auto use_tk_c, auto nearly_c) __attribute__((always_inline)) {
  constexpr bool kUseTk = decltype(use_tk_c)::value;
  constexpr int kNEarly = decltype(nearly_c)::value;
  STRIP_TICK(10);
  __syncthreads();
  STRIP_TICK(11);
#ifdef MLPG_STRIP_TRACE
  tr2 = (long long)__builtin_amdgcn_s_memrealtime();   // level 3 done
#endif

  // ---- level-1 back-substitution and output ----
  const V2 ul = {lds_u[(wv * 2) * 64 + lane], lds_u[(wv * 2 + 1) * 64 + lane]};
  const V2 uo = {lds_u[((wv + 1) * 2) * 64 + lane], lds_u[((wv + 1) * 2 + 1) * 64 + lane]};
  const double sx = lds_u[(kW * 2) * 64 + lane];
  const bool sys_bad = !(sx == sx) || !(uo.x == uo.x) || !(ul.x == ul.x);  // NaN: some pivot of this system failed
  if (MLPG_STRIP_ABLATE < 2) backsub(Pd, P1, P2, rhs, ca, cb, cc, ul, uo);
  STRIP_TICK(12);

  // Verdict.  Strip 0 writes status 0; a strip that met a failing pivot (its own levels 1-2, or a level-3 block in
  // its window: every failure is inside the window of at least the strip that holds it) or a time-out only marks
  // its lanes in the utterance's mask -- the strips far away never learn of it now that level 3 is windowed.
  // verdict_kernel (next launch on the stream) turns the marks into the reference's status and zero columns.
  if (wv == 0) {
    const int to = __builtin_amdgcn_readfirstlane(timed_out);
    const unsigned long long m = to ? ~0ull : __ballot(sys_bad && lane_ok);
    if (m != 0ull && lane == 0) {
      int *line = a.ctrl + (1 + kMaxLists + g) * kCtrlLine;
      __hip_atomic_fetch_or(line + 2, (int)(unsigned)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_or(line + 3, (int)(unsigned)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (to) __hip_atomic_store(line + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#if !defined(MLPG_STRIP_TIMING) && !defined(MLPG_STRIP_TRACE)
    if (r == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + dstat] = 0;
#endif
  }
  const bool zero_out = sys_bad || timed_out;

#ifndef MLPG_STRIP_TIMING
  if (!lane_ok) return;
#endif
  if (!BWD) {
#if MLPG_STRIP_FWD_BUFSTORE
    // trajectory rows by buffer stores: the utterance's descriptor, the row offset in a scalar register, the dim's offset in one
    // vector register (16 stores per chunk; interleaved A/B in round 5: see the switch)
    const __amdgpu_buffer_rsrc_t ors_f = make_rsrc(out_b);
    const unsigned ooff_f = (unsigned)d * (unsigned)sizeof(TOUT), ldo_b = (unsigned)ldo * (unsigned)sizeof(TOUT);
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      const int t = f0 + i;
      if (t < Tmax) st_row(ors_f, (unsigned)t * ldo_b, ooff_f, (t < T && !zero_out) ? (TOUT)rhs[i] : (TOUT)0);
    }
#else
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      const int t = f0 + i;
#if MLPG_STRIP_NT_STORES
      if (t < Tmax) __builtin_nontemporal_store((t < T && !zero_out) ? (TOUT)rhs[i] : (TOUT)0, &out_b[(size_t)t * ldo + d]);
#else
      if (t < Tmax) out_b[(size_t)t * ldo + d] = (t < T && !zero_out) ? (TOUT)rhs[i] : (TOUT)0;
#endif
    }
#endif
  } else {
    // grad[t, w*sd+d] = tau_w[t] * (cm x[t-1] + c0 x[t] + cp x[t+1])  (paramgen/_mlpg.py:202-281).  The row whose
    // right neighbour lives in the next chunk is written by that chunk: this wavefront writes rows f0-1 .. f0+14,
    // and row f0+15 only if it is the utterance's last frame or padding.
    // The variances are read again here (the 51 reciprocals of the assembly are not kept).  All of a window's 17
    // loads are issued before its stores, and the next window's before this window's arithmetic: written as
    // load -> reciprocal -> store per row, the compiler cannot move a load above the preceding store (the two
    // pointers may alias) and the epilogue becomes 51 round trips.
    const unsigned ldi_bytes = (unsigned)ldi * (unsigned)sizeof(TIN), win_bytes = (unsigned)sd * (unsigned)sizeof(TIN);
    auto load_w = [&](TIN (&v)[kM + 1], const int w) __attribute__((always_inline)) {
      if (VM != MLPG_HIP_VAR_FRAME) return;
#pragma unroll
      for (int i = -1; i < kM; ++i) {
        int t = f0 + i;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);  // in bounds; a row that is not live is not used
        v[i + 1] = ld_row<TIN>(vrs, (unsigned)t * ldi_bytes + (unsigned)w * win_bytes, loff);
      }
    };
    auto emit_w = [&](const TIN (&v)[kM + 1], const int w) __attribute__((always_inline)) {
      const double cm = a.wc[w][0], c0 = a.wc[w][1], cp = a.wc[w][2];
      double tau_glob = 1.0;
      if (VM == MLPG_HIP_VAR_GLOBAL) tau_glob = tau_of<TIN>(vglob[w * sd]);
      TOUT *ow = out_b + (size_t)w * sd + d;
#pragma unroll
      for (int i = -1; i < kM; ++i) {
        const int t = f0 + i;
        if (t < 0 || t >= Tmax) continue;
        if (t >= T) {
          if (i >= 0) ow[(size_t)t * ldo] = (TOUT)0;
          continue;
        }
        if (i == -1 && f0 >= T) continue;          // (cannot happen in an active chunk; keeps the rule explicit)
        if (i == kM - 1 && t != T - 1) continue;    // the next chunk writes it
        const bool lv = w ? (mw != 0 && t >= mw && t < T - mw) : true;
        double tau = 0.0;
        if (lv) tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(v[i + 1]) : tau_glob;
        const double xm = (i == -1) ? ul.x : ((i == 0) ? ul.y : rhs[i > 0 ? i - 1 : 0]);
        const double x0 = (i == -1) ? ul.y : rhs[i >= 0 ? i : 0];
        const double xp = (i == kM - 1) ? 0.0 : rhs[i + 1];
        const double gval = tau * (cm * xm + c0 * x0 + cp * xp);
        ow[(size_t)t * ldo] = zero_out ? (TOUT)0 : (TOUT)gval;
      }
    };
    if (MLPG_STRIP_BWD_FRAME_MAJOR && NW3) {
      // Frame-major epilogue (round 3): per frame three reciprocals and THREE ADJACENT 480-byte stores -- the wavefront writes its 17 gradient rows as one contiguous 24 KB run in
      // address order (window-major, the three blocks of a row were written 16 rows of stores apart).
      // All 51 variance loads are issued before the first store: loads and stores share one in-order counter on this
      // chip, so a load queued behind stores waits for the stores' acknowledgements (a ring of loads refilled between
      // the stores ran at one store latency per six frames: 19 k cycles for this epilogue, as long as level 1).
      double tg[3] = {1.0, 1.0, 1.0};
      if (VM == MLPG_HIP_VAR_GLOBAL) {
#pragma unroll
        for (int w = 0; w < 3; ++w) tg[w] = tau_of<TIN>(vglob[w * sd]);
      }
      const __amdgpu_buffer_rsrc_t ors_e = make_rsrc(out_b);
      const unsigned ooff_e = (unsigned)d * (unsigned)sizeof(TOUT);
      (void)ors_e; (void)ooff_e;
      // the three coefficients per window (entries 0 .. 2 of wc[w]), fetched here (see karg_f64x6)
      double we[3][3];
      if (MLPG_STRIP_BWD_KARG) {
        karg_f64x3<kKargWc + 0 * 72>(we[0]);
        karg_f64x3<kKargWc + 1 * 72>(we[1]);
        karg_f64x3<kKargWc + 2 * 72>(we[2]);
      } else {
#pragma unroll
        for (int w = 0; w < 3; ++w) { we[w][0] = a.wc[w][0]; we[w][1] = a.wc[w][1]; we[w][2] = a.wc[w][2]; }
      }
      auto emitf = [&](const int sl) __attribute__((always_inline)) {
        const int i = sl - 1;
        const int t = f0 + i;
        if (t < 0 || t >= Tmax) return;
        TOUT *orow = out_b + (size_t)t * ldo + d;
        unsigned row_off = (unsigned)t * (unsigned)ldo * (unsigned)sizeof(TOUT);
        if (MLPG_STRIP_BWD_BUFSTORE) asm volatile("" : "+s"(row_off));  // (not one of 17 loop invariants held in scalar registers)
        auto put = [&](const int w, const TOUT val) __attribute__((always_inline)) {
#if MLPG_STRIP_BWD_BUFSTORE
          st_row(ors_e, row_off + (unsigned)w * (unsigned)sd * (unsigned)sizeof(TOUT), ooff_e, val);
          return;
#endif
#if MLPG_STRIP_NT_STORES
          __builtin_nontemporal_store(val, orow + (size_t)w * sd);
#else
          orow[(size_t)w * sd] = val;
#endif
        };
        if (t >= T) {
          if (i >= 0) { put(0, (TOUT)0); put(1, (TOUT)0); put(2, (TOUT)0); }
          return;
        }
        if (i == -1 && f0 >= T) return;
        if (i == kM - 1 && t != T - 1) return;  // the next chunk writes it
        const double xm = (i == -1) ? ul.x : ((i == 0) ? ul.y : rhs[i > 0 ? i - 1 : 0]);
        const double x0 = (i == -1) ? ul.y : rhs[i >= 0 ? i : 0];
        const double xp = (i == kM - 1) ? 0.0 : rhs[i + 1];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const bool lv = w ? (mw != 0 && t >= mw && t < T - mw) : true;
          double tau = 0.0;
          if (kUseTk) {
            // dead frames were kept as 0.  The opaque copy keeps the widening HERE: left to itself the optimiser widens all
            // 51 values right behind the assembly and carries them as doubles (102 registers) across levels 2 and 3
            float tf = tk[sl][w];
            asm volatile("" : "+v"(tf));
            tau = (double)tf;
          }
          else if (lv) tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(tv[sl][w]) : tg[w];
          const double gval = tau * (we[w][0] * xm + we[w][1] * x0 + we[w][2] * xp);
          put(w, zero_out ? (TOUT)0 : (TOUT)gval);
        }
      };
      if (!kUseTk) {
#pragma unroll
        for (int sl = kNEarly; sl < kEpi; ++sl) ldf(tv[sl], sl - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#define STRIP_EPI(S)                                                        \
      if ((S) < kEpi) {                                                       \
      emitf((S));                                                             \
      __builtin_amdgcn_sched_barrier(0);                                      \
      }
      STRIP_EPI(0) STRIP_EPI(1) STRIP_EPI(2) STRIP_EPI(3) STRIP_EPI(4) STRIP_EPI(5) STRIP_EPI(6) STRIP_EPI(7) STRIP_EPI(8)
      STRIP_EPI(9) STRIP_EPI(10) STRIP_EPI(11) STRIP_EPI(12) STRIP_EPI(13) STRIP_EPI(14) STRIP_EPI(15) STRIP_EPI(16)
#undef STRIP_EPI
    } else {
    TIN tvA[kM + 1], tvB[kM + 1];
      load_w(tvA, 0);
      for (int w = 0; w < nw; w += 2) {
        if (w + 1 < nw) load_w(tvB, w + 1);
        __builtin_amdgcn_sched_barrier(0);
        emit_w(tvA, w);
        __builtin_amdgcn_sched_barrier(0);
        if (w + 1 < nw) {
          if (w + 2 < nw) load_w(tvA, w + 2);
          __builtin_amdgcn_sched_barrier(0);
          emit_w(tvB, w + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    }
#ifdef MLPG_STRIP_TIMING
  // profiling build only: phase cycle counts of wavefront 0 of strips 0..15 of utterances 0..7 overwrite the head of
  // the status array (run bench.py --no-check with MLPG_DUMP_STATUS=1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STRIP_TICK(13);
  if (wv == 0 && lane == 0 && p.status && b < 8 && r < 16 && dg == 0)
  {
    for (int k = 0; k < 16; ++k) p.status[(b * 16 + r) * 16 + k] = (int)tq[k];
  }
#endif
#ifdef MLPG_STRIP_TRACE
  if (wv == 0 && lane == 0 && p.status && (g * R + r) * 4 + 3 < p.B * p.ld_status) {
    int *tp = p.status + (g * R + r) * 4;
    tp[0] = (int)(tr0 & 0x3FFFFFFF); tp[1] = (int)(tra - tr0) | ((int)(tr1 - tr0) << 16); tp[2] = (int)(tr2 - tr0);
    tp[3] = (int)((long long)__builtin_amdgcn_s_memrealtime() - tr0) | (xcd << 24);
  }
#endif
  };  // tail
  if (wv == 0) {
    if (kParkOn) {
#pragma unroll
      for (int i = 0; i < kN; ++i) {
        lds_park[i * 64 + lane] = rhs[i];
        lds_park[(kN + i) * 64 + lane] = P2[i];
      }
    }
    S2 E_s = {1.0, 0.0, 1.0};
    V2 g_s = {0.0, 0.0};
    auto R_ = [&](int j, int k) __attribute__((always_inline)) { return lds_rec[(j * kRec + k) * 64 + lane]; };  // level-1 records
    S2 E = {R_(0, rD11), R_(0, rD12), R_(0, rD22)};
    V2 gg = {R_(0, rF1), R_(0, rF2)};
    M2 V = {R_(0, rL11), R_(0, rL12), R_(0, rL21), R_(0, rL22)};
    if (r == 0) V = {0.0, 0.0, 0.0, 0.0};
    S2 Ts = {R_(0, rT00), R_(0, rT01), R_(0, rT11)};
    V2 hs = {R_(0, rH0), R_(0, rH1)};
    if (kW > 1) {
      E = sub(E, S2{R_(1, rT00), R_(1, rT01), R_(1, rT11)});
      gg = sub(gg, V2{R_(1, rH0), R_(1, rH1)});
    }
    bool bad2 = false;
#pragma unroll
    for (int j = 0; j + 1 < kW; ++j) {
      const S2 Einv = sym_inv(E, bad2);
      const M2 L = {R_(j + 1, rL11), R_(j + 1, rL12), R_(j + 1, rL21), R_(j + 1, rL22)};
      const M2 Mn = mul_ms(L, Einv);
      const M2 EV = mul_sm(Einv, V);
      const V2 c = mul_sv(Einv, gg);
      Ts = add(Ts, mul_mtm_sym(V, EV));
      hs = add(hs, mul_mtv(V, c));
      double *f = lds_fac + (size_t)j * kFac * 64 + lane;
      f[0 * 64] = c.x; f[1 * 64] = c.y;
      f[2 * 64] = EV.a; f[3 * 64] = EV.b; f[4 * 64] = EV.c; f[5 * 64] = EV.d;
      f[6 * 64] = Mn.a; f[7 * 64] = Mn.b; f[8 * 64] = Mn.c; f[9 * 64] = Mn.d;
      S2 Dn = {R_(j + 1, rD11), R_(j + 1, rD12), R_(j + 1, rD22)};
      V2 Fn = {R_(j + 1, rF1), R_(j + 1, rF2)};
      if (j + 2 < kW) {
        Dn = sub(Dn, S2{R_(j + 2, rT00), R_(j + 2, rT01), R_(j + 2, rT11)});
        Fn = sub(Fn, V2{R_(j + 2, rH0), R_(j + 2, rH1)});
      }
      E = sub(Dn, mul_mmt_sym(Mn, L));
      gg = sub(Fn, mul_mv(Mn, gg));
      V = neg(mul_mm(Mn, V));
      // the chain is sequential: keep the scheduler from hoisting every later record read to the top (this
      // wavefront holds its chunk's factor in 112 registers meanwhile)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bad2) E.a = __builtin_nan("");
    E_s = E;
    g_s = gg;
    STRIP_TICK(4);
    int route = kLocal;  // strips per side of the level-3 window (2, 4, 8 or 16); 0: the whole utterance
    if (xwg) {
      // publish the strip's record, then announce it
      double *rp = a.rec + ((size_t)g * R + r) * (kRec * 64) + lane;
      st_agent(rp + 0 * 64, E.a); st_agent(rp + 1 * 64, E.b); st_agent(rp + 2 * 64, E.c);
      st_agent(rp + 3 * 64, gg.x); st_agent(rp + 4 * 64, gg.y);
      st_agent(rp + 5 * 64, V.a); st_agent(rp + 6 * 64, V.b); st_agent(rp + 7 * 64, V.c); st_agent(rp + 8 * 64, V.d);
      st_agent(rp + 9 * 64, Ts.a); st_agent(rp + 10 * 64, Ts.b); st_agent(rp + 11 * 64, Ts.c);
      st_agent(rp + 12 * 64, hs.x); st_agent(rp + 13 * 64, hs.y);
      // Route, from this strip's own data alone (so that the choice -- and with it every bit of the result -- never
      // depends on timing): its own transfer factor t = 2 max|E^-1 V| is one of the factors of the window's damping
      // bound (k per side in a window of k strips per side).  t <= kDampTol^(1/2): the narrow window; otherwise the
      // smallest k in {4, 8, 16} with t^k <~ kDampTol; beyond that, or if such a window spans the utterance anyway
      // (variances whose dynamic features are much tighter than the static ones couple strips over hundreds of
      // frames), the strip waits for the whole utterance right away.  The bound still decides whether a window's
      // result is accepted.
      {
        bool badr = false;
        const double t_own = 2.0 * amax4(mul_sm(sym_inv(E, badr), V));
        // k strips per side need t^k <~ kDampTol: 2 (t <= 1e-11), 4 (3e-6), 8 (1.8e-3), 16 (4.2e-2); 0 = whole utterance
        auto any_over = [&](const double tol) { return __ballot(lane_ok && !(t_own <= tol)) != 0ull; };  // NaN counts
        route = !any_over(MLPG_STRIP_ROUTE1_TOL) ? 1 : !any_over(kRouteTol) ? kLocal : !any_over(3e-6) ? 4 : !any_over(1.8e-3) ? 8 : !any_over(4.2e-2) ? 16 : 0;
        if (2 * route + 1 >= Ract) route = route > kLocal ? 0 : route;  // a window as wide as the utterance: sweep it all
        if (lane == 0) lds_misc[3] = route;
      }
      const int kwin = route ? route : kLocal;
      if (route && route <= kLocal && 2 * route + 1 <= kStage) {
        // the narrow windows' sweep takes this strip's own record from here (the last place of the staged order),
        // not from HBM.  (lds_rec, the same bytes, has been consumed by level 2.)
        const Window w = local_window(r, Ract, route);
        const Order o = make_order(r, w.lo, w.hiE);
        const double own[kRec] = {E.a, E.b, E.c, gg.x, gg.y, V.a, V.b, V.c, V.d, Ts.a, Ts.b, Ts.c, hs.x, hs.y};
#pragma unroll
        for (int k = 0; k < kRec; ++k) lds_stage[((o.npos - 1) * kRec + k) * 64 + lane] = own[k];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STRIP_TICK(5);
      // announce: the utterance's arrival counter (for the full sweep) and this strip's own flag (for the neighbours)
      int *cnt = a.ctrl + (1 + kMaxLists + g) * kCtrlLine;
      int *flags = a.ctrl + (1 + kMaxLists + a.nsg) * kCtrlLine + (size_t)g * flag_pitch(R);
      if (lane == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flags + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // wait for the strips of the local window only (lane l polls the flag of strip wlo + l), or for the whole utterance
      {
        const Window w = local_window(r, Ract, kwin);
        int spins = 0, ok = 1;
        for (;;) {
          int f = 1;
          if (route == 0) f = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= Ract;
          else if (lane <= w.hiE - w.lo) f = __hip_atomic_load(flags + w.lo + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__ballot(f == 0) == 0ull) break;
          __builtin_amdgcn_s_sleep(MLPG_STRIP_POLL_SLEEP);
          if (++spins > kSpinLimit) { ok = 0; break; }
        }
        if (lane == 0) {
          if (!ok) atomicAdd(a.ctrl, 1);
          lds_misc[1] = ok;
        }
      }
      STRIP_TICK(6);
    }
    __syncthreads();  // (S2) every wavefront: the utterance's strips have all arrived (or the wait timed out)
    STRIP_TICK(7);
#ifdef MLPG_STRIP_TRACE
    tr1 = (long long)__builtin_amdgcn_s_memrealtime();   // all strips of the utterance have arrived
#endif
    // ---- level 3: every strip sweeps all records of its utterance ----
    V2 sig = {0.0, 0.0}, sprev = {0.0, 0.0};  // solution on this strip's last separator / the previous strip's
    if (xwg) {
      timed_out = !__builtin_amdgcn_readfirstlane(lds_misc[1]);
      // Two sweeps over the records of rows lo .. hiE, no factor stored: top-down over rows lo .. r-1 (row j is
      // finalised when row j+1 is at hand: A_j = E_j - T_{j+1} - Mn_j V_j^T with Mn_j = V_j A_{j-1}^-1, a_j likewise),
      // bottom-up over rows hiE .. r+1 (the Schur complement (S, s) of the rows below row j: B_j = E_j - T_{j+1} - S,
      // S' = V_j^T B_j^-1 V_j), and the 2-block system of rows r-1 and r in the middle.  The two eliminations are
      // independent chains: the records are staged as (top row, bottom row) pairs moving inwards (Order), a pair is
      // one straight-line block, row r comes last.
      // With lo = 0, hiE = Ract-1 this is the exact solve.  A narrower window clamps the separators just outside it
      // to zero: separator lo-1 (record lo holds strip lo's interior and its coupling V_lo to that separator) and,
      // with `edge`, separator hiE (record hiE is read for T, h -- that strip's interior -- and its coupling V only).
      // What this ignores is exactly V_lo u_{lo-1} on the window's first row and V_hiE^T u_hiE on its last, and
      // they reach rows r-1, r through the transfer matrices of the two eliminations:
      //     top:     |du_{r-1}| <= prod_{j=lo}^{r-1} |A_j^-1 V_j| |u_{lo-1}|,    du_r: one more factor |S_r^-1 V_r|
      //     bottom:  |du_r| <= |S_r^-1 V_{r+1}^T| prod_{j=r+1}^{hiE-1} |B_j^-1 V_{j+1}^T| |u_hiE|,
      //              du_{r-1}: one more factor |A_{r-1}^-1 V_r^T|
      // (|.| of a 2x2 block <= twice its largest entry).  `damp` is the larger product.  Separators 64 frames apart
      // are coupled by ~1e-20 for ordinary variances, so the window reproduces the exact solve to the last bit;
      // where the bound says otherwise (long spans of vanishing static precision) the sweep is repeated over the
      // whole utterance.  tools/strip_model.py: utterance_solve_two_sided.
      double damp = 0.0;
      auto sweep = [&](const int lo, const int hiE, const int edge) __attribute__((always_inline)) {
        const Order o = make_order(r, lo, hiE);
        S2 Ainv = {0.0, 0.0, 0.0};
        V2 av = {0.0, 0.0};
        M2 Mn = {0.0, 0.0, 0.0, 0.0};
        S2 Sb = {0.0, 0.0, 0.0}, Tn = {0.0, 0.0, 0.0};
        V2 sb = {0.0, 0.0}, hn = {0.0, 0.0};
        M2 Vn = {0.0, 0.0, 0.0, 0.0};  // coupling of the row below to the current one
        double dt = lo > 0 ? 1.0 : 0.0, db = edge ? 1.0 : 0.0;
        bool bad3 = false;
        // the pending row of the top-down sweep (E, g, V of row j until row j+1 arrives)
        S2 Ej = {0.0, 0.0, 0.0};
        V2 gj = {0.0, 0.0};
        M2 Vj = {0.0, 0.0, 0.0, 0.0};
        struct Rec { S2 E; V2 g; M2 V; S2 T; V2 h; };
        auto rd = [&](const int q) __attribute__((always_inline)) {
          double c[kRec];
#pragma unroll
          for (int kk = 0; kk < kRec; ++kk) c[kk] = lds_stage[(q * kRec + kk) * 64 + lane];
          return Rec{{c[0], c[1], c[2]}, {c[3], c[4]}, {c[5], c[6], c[7], c[8]}, {c[9], c[10], c[11]}, {c[12], c[13]}};
        };
        // top-down: the next row has arrived; the pending one is finalised (Mn = 0 for the window's first row)
        auto top_finalize = [&](const Rec &k) __attribute__((always_inline)) {
          const S2 A = sub(sub(Ej, k.T), mul_mmt_sym(Mn, Vj));
          const V2 aa = sub(sub(gj, k.h), mul_mv(Mn, av));
          Ainv = sym_inv(A, bad3);
          av = aa;
          Mn = mul_ms(k.V, Ainv);  // V_{j+1} A_j^-1
          dt *= 2.0 * amax4(mul_sm(Ainv, Vj));
        };
        auto top_pend = [&](const Rec &k) __attribute__((always_inline)) { Ej = k.E; gj = k.g; Vj = k.V; };
        // bottom-up: row j > r; the clamped edge below a window contributes its strip's interior and its coupling only
        auto bot_edge = [&](const Rec &k) __attribute__((always_inline)) { Tn = k.T; hn = k.h; Vn = k.V; };
        auto bot_row = [&](const Rec &k) __attribute__((always_inline)) {
          const S2 B = sub(sub(k.E, Tn), Sb);
          const V2 bv = sub(sub(k.g, hn), sb);
          const S2 Binv = sym_inv(B, bad3);
          db *= 2.0 * amax4(mul_smt(Binv, Vn));
          const M2 Wm = mul_sm(Binv, k.V);
          Sb = mul_mtm_sym(k.V, Wm);
          sb = mul_mtv(k.V, mul_sv(Binv, bv));
          Tn = k.T; hn = k.h; Vn = k.V;
        };
        for (int p0 = 0; p0 < o.npos; p0 += kStage) {
          const int kn = o.npos - p0 < kStage ? o.npos - p0 : kStage;
          __syncthreads();  // this batch is in LDS
          STRIP_TICK(8);
          if (!timed_out) {
#ifndef MLPG_L3_NOPRIO
            __builtin_amdgcn_s_setprio(2);  // the whole workgroup waits for this chain
#endif
            int q = 0;
            while (q < kn) {
              const int pos = p0 + q;
              if (pos < 2 * o.m) {
                // a pair: top row lo + idx and bottom row hiE - idx (kStage is even: a pair never straddles batches)
                const int idx = pos >> 1;
                const Rec kt = rd(q), kb = rd(q + 1);
                if (idx == 0) {
                  top_pend(kt);
                  if (edge) bot_edge(kb);
                  else bot_row(kb);
                } else {
                  top_finalize(kt);  // two independent chains in one block
                  bot_row(kb);
                  top_pend(kt);
                }
                q += 2;
              } else if (pos < o.nt + o.nb) {
                const int idx = o.m + (pos - 2 * o.m);
                const Rec k = rd(q);
                if (o.nt > o.nb) {
                  if (idx > 0) top_finalize(k);
                  top_pend(k);
                } else {
                  if (idx == 0 && edge) bot_edge(k);
                  else bot_row(k);
                }
                q += 1;
              } else {
                // row r: the last top row is finalised, then the 2-block system of rows r-1, r
                const Rec k = rd(q);
                if (o.nt > 0) top_finalize(k);  // Mn = V_r A_{r-1}^-1
                S2 B = sub(sub(k.E, Tn), Sb);
                V2 bv = sub(sub(k.g, hn), sb);
                if (o.nt > 0) {
                  B = sub(B, mul_mmt_sym(Mn, k.V));
                  bv = sub(bv, mul_mv(Mn, av));
                }
                const S2 Binv = sym_inv(B, bad3);
                sig = mul_sv(Binv, bv);
                sprev = {0.0, 0.0};
                db *= 2.0 * amax4(mul_smt(Binv, Vn));
                if (o.nt > 0) {
                  sprev = sub(mul_sv(Ainv, av), mul_mtv(Mn, sig));  // A^-1 (a - V_r^T sigma_r)
                  dt *= __builtin_fmax(1.0, 2.0 * amax4(mul_sm(Binv, k.V)));
                  db *= __builtin_fmax(1.0, 2.0 * amax4(mul_smt(Ainv, k.V)));
                }
                if (bad3) sig.x = __builtin_nan("");
                q += 1;
              }
            }
            __builtin_amdgcn_s_setprio(0);
          }
          STRIP_TICK(9);
          __syncthreads();  // done with this batch
        }
        damp = dt > db ? dt : db;
      };
      if (route == 0) {
        sweep(0, Ract - 1, 0);
      } else {
        // The ladder (round 5): the window the strip routed itself to; if its bound is not small enough, a rejected 3-strip
        // window is followed by the 5-strip one (its records from HBM this time), any other by the whole utterance.  Every
        // decision depends on the strip's and its neighbours' DATA only, never on timing.
        int cur = route;
        int *cnt = a.ctrl + (1 + kMaxLists + g) * kCtrlLine;
        int *flags = a.ctrl + (1 + kMaxLists + a.nsg) * kCtrlLine + (size_t)g * flag_pitch(R);
        for (;;) {
          const Window w = local_window(r, Ract, cur);
          const bool full_range = w.lo == 0 && !w.edge;
          sweep(w.lo, w.hiE, w.edge);
          // accept the windowed result only if every system of the strip is damped far below the rounding level and
          // met no failing pivot (those are re-examined on the whole utterance, so that the verdict never depends on
          // the window)
          const double tolw = cur == 1 ? kDamp1Tol : kDampTol;
          const bool lane_fine = !lane_ok || full_range || (damp < tolw && sig.x == sig.x);
          const int accept = timed_out || __ballot(!lane_fine) == 0ull;
          // 0: done; 1: the 5-strip window next (after a 3-strip one that does not already reach that far); 2: the whole utterance
          const int next = accept ? 0 : ((cur == 1 && kLocal > 1) ? 1 : 2);
          int ok = 1;
          if (next == 1) {
            // the two strips the wider window adds may not have arrived yet (lane l polls the flag of strip wlo + l)
            const Window w2 = local_window(r, Ract, kLocal);
            int spins = 0;
            for (;;) {
              int f = 1;
              if (lane <= w2.hiE - w2.lo) f = __hip_atomic_load(flags + w2.lo + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (__ballot(f == 0) == 0ull) break;
              __builtin_amdgcn_s_sleep(MLPG_STRIP_POLL_SLEEP);
              if (++spins > kSpinLimit) { ok = 0; break; }
            }
          } else if (next == 2) {
            // the whole utterance is needed: wait for all of its strips
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < Ract) {
              __builtin_amdgcn_s_sleep(32);
              if (++spins > kSpinLimit) { ok = 0; break; }
            }
          }
          if (lane == 0) {
            if (!ok) atomicAdd(a.ctrl, 1);
            if (next) lds_misc[1] = ok;
            lds_misc[2] = next;
          }
          __syncthreads();  // (S2b) the stagers learn the decision
          if (next == 0) break;
          timed_out = !ok;
          if (next == 2) {
            sweep(0, Ract - 1, 0);
            break;
          }
          cur = kLocal;
        }
      }
    } else {
      bool bad3 = false;
      const S2 Ai = sym_inv(E_s, bad3);
      sig = mul_sv(Ai, g_s);
      if (bad3) sig.x = __builtin_nan("");
    }

    // wavefront 0's share of the early variance rows: only now has it registers to receive them (levels 2 and 3 are done);
    // they travel during the two back-substitutions
    if (kEarly0 > 0) early_issue0();
    // ---- back-substitution of level 2 ----
    double *up = lds_u + lane;
    up[0] = sprev.x; up[64] = sprev.y;
    up[(kW * 2) * 64] = sig.x; up[(kW * 2 + 1) * 64] = sig.y;
    V2 un = sig;
#pragma unroll
    for (int j = kW - 2; j >= 0; --j) {
      const double *f = lds_fac + (size_t)j * kFac * 64 + lane;
      const V2 c = {f[0 * 64], f[1 * 64]};
      const M2 EV = {f[2 * 64], f[3 * 64], f[4 * 64], f[5 * 64]};
      const M2 Mn = {f[6 * 64], f[7 * 64], f[8 * 64], f[9 * 64]};
      const V2 uj = sub(sub(c, mul_mv(EV, sprev)), mul_mtv(Mn, un));
      up[((j + 1) * 2) * 64] = uj.x; up[((j + 1) * 2 + 1) * 64] = uj.y;
      un = uj;
    }
    if (kParkOn) {
#pragma unroll
      for (int i = 0; i < kN; ++i) {
        rhs[i] = lds_park[i * 64 + lane];
        P2[i] = lds_park[(kN + i) * 64 + lane];
      }
    }
    if (kSplitTail) tail(std::integral_constant<bool, kKeepTau && MLPG_STRIP_BWD_KEEP0>{}, std::integral_constant<int, kEarly0>{});
  } else {
    __syncthreads();  // (S2)
    // wavefronts 1..3: stage the records (kStage rows per batch) through LDS; the loads of batch k+1 are issued into
    // registers before wavefront 0 starts on batch k, so only the first batch's memory latency is exposed
    if (xwg) {
      timed_out = !__builtin_amdgcn_readfirstlane(lds_misc[1]);
      constexpr int kSlots = (kStage + kW - 2) / (kW - 1);  // rows per stager per batch
      double sv[kSlots][kRec];
      bool own_slot[kSlots];
      // staged record number p of the window [lo, hi]: row_of(make_order(r, lo, hi), p)
      bool skip_own = true;  // windowed sweep: wavefront 0 has put this strip's own record into its LDS slots
      auto stage_load = [&](int lo, int hi, int p0, int kn) __attribute__((always_inline)) {
        const Order o = make_order(r, lo, hi);
#pragma unroll
        for (int sl = 0; sl < kSlots; ++sl) {
          const int q = (wv - 1) + sl * (kW - 1);
          const int pos = p0 + q;
          const int row_ = row_of(o, pos);
          own_slot[sl] = skip_own && pos == o.npos - 1;
          if (q < kn && !timed_out && !own_slot[sl]) {
            const int row = row_;
            const double *rp = a.rec + ((size_t)g * R + row) * (kRec * 64) + lane;
#pragma unroll
            for (int k = 0; k < kRec; ++k) sv[sl][k] = ld_agent(rp + k * 64);
          }
        }
      };
      auto stage = [&](const int lo, const int hi) __attribute__((always_inline)) {
        const int npos = hi - lo + 1;
        stage_load(lo, hi, 0, npos < kStage ? npos : kStage);
        for (int p0 = 0; p0 < npos; p0 += kStage) {
          const int kn = npos - p0 < kStage ? npos - p0 : kStage;
#pragma unroll
          for (int sl = 0; sl < kSlots; ++sl) {
            const int q = (wv - 1) + sl * (kW - 1);
            if (q < kn && !timed_out && !own_slot[sl]) {
#pragma unroll
              for (int k = 0; k < kRec; ++k) lds_stage[(q * kRec + k) * 64 + lane] = sv[sl][k];
            }
          }
          __syncthreads();  // this batch is in LDS
          if (p0 + kStage < npos) stage_load(lo, hi, p0 + kStage, npos - p0 - kStage < kStage ? npos - p0 - kStage : kStage);
          __syncthreads();  // wavefront 0 has read this batch
        }
      };
      const int route = __builtin_amdgcn_readfirstlane(lds_misc[3]);  // wavefront 0's choice (see there)
      if (route == 0) {
        skip_own = false;
        stage(0, Ract - 1);
      } else {
        skip_own = route <= kLocal && 2 * route + 1 <= kStage;  // only the narrow (single-batch) windows have this strip's own record in LDS already
        int cur = route;
        for (;;) {  // wavefront 0's ladder, mirrored
          const Window w = local_window(r, Ract, cur);
          stage(w.lo, w.hiE);
          __syncthreads();  // (S2b) wavefront 0's decision: is the window enough?
          const int next = __builtin_amdgcn_readfirstlane(lds_misc[2]);
          if (next == 0) break;
          timed_out = !__builtin_amdgcn_readfirstlane(lds_misc[1]);
          skip_own = false;  // from here on every row is staged from HBM
          if (next == 2) {
            stage(0, Ract - 1);
            break;
          }
          cur = kLocal;
        }
      }
    }
    if (kSplitTail) tail(std::integral_constant<bool, kKeepTau>{}, std::integral_constant<int, kEarly>{});
  }
  if (!kSplitTail) tail(std::false_type{}, std::integral_constant<int, kEarly>{});
  };  // body

  // (Starting the second workgroup of each CU half an item late was measured in round 2 -- no gain -- and removed in round 5; round 6's
  // trace shows why: whatever the start, the workgroups of an utterance's strips re-align within two items.  See the start ramp in
  // launch_impl for the part of a staggered start that does pay.)
  bool first_item = true;
  const long long t_launch = (long long)__builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < a.nlists; ++k) {
    const int lst = (xcd + k) % a.nlists;
    // items of this list.  MULTI: the lists are dealt by UTTERANCE (all its dim groups in one list, a full group and
    // the narrow last one alternating) -- by system group, the full groups of 66 = 64 + 2 dims would all land in
    // the even lists and the XCDs behind the odd ones would idle
    const int lim = MULTI ? ((a.nsg / a.ndg - lst + a.nlists - 1) / a.nlists) * a.ndg * R
                          : ((a.nsg * a.nb - lst + a.nlists - 1) / a.nlists) * a.bs;
    int *ticket = a.ctrl + (1 + lst) * kCtrlLine;
    for (;;) {
      if (tid == 0) {
        int tk = lim;  // a plain look first: an exhausted list costs no read-modify-write
        if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lim) tk = atomicAdd(ticket, 1);
        if (first_item && a.stagger > 0 && tk < a.wpl && tk < lim) {
          // the start ramp (round 6): the first-round workgroups of a list start their first item spread over `stagger` ticks in
          // ticket order, so that the launch does not open with every workgroup loading at once (see launch_impl)
          const long long until = t_launch + (long long)tk * a.stagger / a.wpl;
          while ((long long)__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
        }
        lds_misc[0] = tk;
      }
      first_item = false;
      __syncthreads();
      const int tk = __builtin_amdgcn_readfirstlane(lds_misc[0]);
      if (tk >= lim) break;
#ifdef MLPG_STRIP_TIMING
      t_prev = (long long)__builtin_readcyclecounter();
      for (int q = 0; q < 16; ++q) tq[q] = 0;
#endif
      int g_it, r_it;
      ticket_item<MULTI>(a, tk, lst, g_it, r_it);
      if (r_it < R) body(g_it, r_it);  // (the last block of a long utterance may be short: tickets past its end are nobody's)
      __syncthreads();
    }
    __syncthreads();
  }
}

// ---- verdict ------------------------------------------------------------------------------------
// One wavefront per system group, launched after strip_kernel on the same stream.  Marked lanes (see above) get the
// reference's verdict: status = natural-order first failing pivot (-2 if that scan finds none: the blocked
// elimination broke down on a numerically singular system; -1 after a time-out) and an all-zero output column,
// exactly what the other kernels deliver.  It also re-zeroes the control words for the next launch.
template <typename TIN, typename TOUT, bool BWD, bool MULTI = false, bool TR = false>
__global__ void __launch_bounds__(256) verdict_kernel(const Problem p, const WinSet ws, const Args a) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= a.nsg) return;
  int *line = a.ctrl + (1 + kMaxLists + g) * kCtrlLine;
  const unsigned long long m = (unsigned long long)(unsigned)line[2] | ((unsigned long long)(unsigned)line[3] << 32);
  const int timed_out = line[4];
  // leave every control word as the next launch needs it: zero (launch_strip then skips its memset)
  __builtin_amdgcn_wave_barrier();
  if (lane < 5) line[lane] = 0;
  for (int i = lane; i < flag_pitch(a.R); i += 64) a.ctrl[(1 + kMaxLists + a.nsg) * kCtrlLine + (size_t)g * flag_pitch(a.R) + i] = 0;
  if (g == 0)
    for (int i = lane; i < (1 + kMaxLists) * kCtrlLine; i += 64) a.ctrl[i] = 0;
  if (m == 0ull) return;
  const int b = TR ? g * a.sm.tr_u : g / a.ndg, dg = TR ? 0 : g - b * a.ndg;
  const int d0 = dg * a.dgw;
  const int sd_all = MULTI ? a.sm.total : p.sd;
  const int nd = TR ? (a.sm.tr_B - b < a.sm.tr_u ? a.sm.tr_B - b : a.sm.tr_u) * a.sm.tr_nd
                    : (sd_all - d0 < a.dgw ? sd_all - d0 : a.dgw);
  if (lane >= nd || !((m >> lane) & 1ull)) return;
  int d = d0 + lane, dstat = d;
  const int Tmax = p.Tmax;
  int T = p.lengths ? p.lengths[TR ? b + d / a.sm.tr_nd : b] : Tmax;  // (TR: this lane's utterance)
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  int status = -1;
  if (MULTI) {
    // the dim's own stream as a problem of its own: column slices of the parent arrays
    const LaneStream ls = TR ? lane_stream_tr(a.sm, d) : lane_stream(a.sm, d);
    Problem q = p;
    q.sd = ls.sd;
    q.D = ws.nw * ls.sd;
    // make_view addresses column (w * sd + dloc) of q.mean: shift the bases so that dloc = 0 is this dim
    q.mean = (const TIN *)p.mean + ls.din;
    // (a global (D,) variance vector has no utterance offset)
    q.var = p.var ? (const void *)((const TIN *)p.var + (p.var_mode == MLPG_HIP_VAR_GLOBAL ? ls.dvar : ls.din)) : nullptr;
    if (!timed_out) {
      const SysView<TIN, BWD> view = make_view<TIN, BWD>(q, ws, b, 0, T);
      status = first_bad_pivot<2, TIN, BWD>(view, ws);
      if (status == 0) status = -2;
    }
    dstat = ls.dstat;
    d = ls.dout;
  } else if (!timed_out) {
    const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
    status = first_bad_pivot<2, TIN, BWD>(view, ws);
    if (status == 0) status = -2;
  }
  if (p.status) p.status[(size_t)b * p.ld_status + dstat] = status;
  TOUT *out_b = (TOUT *)p.out + (size_t)b * Tmax * p.ld_out;
  for (int t = 0; t < Tmax; ++t) {
    if (!BWD) out_b[(size_t)t * p.ld_out + d] = (TOUT)0;
    else
      for (int w = 0; w < ws.nw; ++w) out_b[(size_t)t * p.ld_out + w * p.sd + d] = (TOUT)0;
  }
}

// ---- launcher ------------------------------------------------------------------------------------
// Scratch layout for one launch: control words, then the records.
inline size_t ctrl_bytes(int nsg, int R) { return (ctrl_ints(nsg, R) * sizeof(int) + 255) / 256 * 256; }

constexpr int kNotResident = -1000;  // launch_t: the grid cannot hold an utterance's strips; nothing was enqueued

// Workgroups of `kern` (threads, dynamic LDS bytes) that are resident at once on the current device: the occupancy
// the runtime reports x the CU count.  Queried (and the dynamic-LDS attribute set) once per kernel and device.
inline int resident_grid(const void *kern, int threads, size_t lds, int *out) {
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
  MLPG_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MLPG_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  MLPG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
  if (ncu < 1 || per_cu < 1) {
    set_error("strip kernel: occupancy query returned %d workgroups per CU on %d CUs", per_cu, ncu);
    return MLPG_HIP_ERUNTIME;
  }
  std::lock_guard<std::mutex> lk(mu);
  cache.push_back({kern, dev, ncu * per_cu});
  *out = ncu * per_cu;
  return 0;
}

template <typename TIN, typename TOUT, bool BWD, bool MULTI, bool TR = false>
int launch_impl(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, int R, int ndg, int dgw,
                bool zero_ctrl, const StreamMap *smap) {
  Args a;
  memset(&a.sm, 0, sizeof(a.sm));
  if (MULTI) a.sm = *smap;
  const int nsg = TR ? (p.B + smap->tr_u - 1) / smap->tr_u : p.B * ndg;  // (TR: blocks of utterances, one dim group)
  a.ctrl = (int *)scratch_base;
  a.rec = (double *)((char *)scratch_base + ctrl_bytes(nsg, R));
  a.R = R;
  a.ndg = ndg;
  a.dgw = dgw;
  a.nsg = nsg;
  a.one = 1.0;
  for (int w = 0; w < ws.nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *cw = ws.c + ws.off[w];
    const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
    const double v[9] = {cm, c0, cp, c0 * c0, cp * cp, cm * cm, cp * c0, c0 * cm, cp * cm};
    for (int q = 0; q < 9; ++q) a.wc[w][q] = v[q];
  }
  const long nitems = (long)nsg * R;
  auto go = [&](auto kern) -> int {
    // persistent workgroups: as many as can be RESIDENT (asked of the runtime for this kernel and its LDS, once per
    // device), each draws items until the lists are empty
    int resident = 0;
    if (int rc = resident_grid((const void *)kern, kW * 64, kLdsBytes, &resident)) return rc;
    // Co-residency the protocol relies on: a strip may wait for every strip of its utterance (route 0).  One list per
    // XCD (locality: neighbouring strips exchange their records through one L2), and of one utterance a list holds at
    // most cap = half an XCD's share of the grid in consecutive tickets; small launches: one list, and then the whole
    // grid must be able to hold an utterance twice over -- if not, the caller takes another kernel.
    // Longer utterances are dealt to the lists in blocks of consecutive strips that do fit: every list runs through
    // the utterances in the same order, so the oldest unfinished utterance has, in every list, all its strips drawn
    // (a list's first unfinished ticket is one of its strips and the >= 2 * cap workgroups of that XCD hold distinct
    // tickets from there on) -- it completes, then the next one.  Only the block ends talk across XCDs.
    const int cap = resident / (2 * kMaxLists);
    a.nb = 1;
    a.bs = R;
    a.nlists = 1;
    // (an utterance spread over several lists needs every list to have home workgroups: only on a device whose workgroups
    // land on eight XCDs in equal shares -- strip_xcd_lists_ok(), probed once per device; otherwise such launches keep ONE
    // list, which holds whole utterances and depends on no placement)
    if (nitems >= 512 && cap >= 1 && (R <= cap || (!MULTI && strip_xcd_lists_ok(st)))) {
      a.nlists = kMaxLists;
      a.nb = (R + cap - 1) / cap;
      a.bs = (R + a.nb - 1) / a.nb;
    }
    if (nitems > resident && R > resident / 2) return kNotResident;
    if (zero_ctrl) MLPG_HIP_CHECK(hipMemsetAsync(a.ctrl, 0, ctrl_ints(nsg, R) * sizeof(int), st));
    long grid = nitems < resident ? nitems : resident;
    {
      // measurement only (MLPG_STRIP_GRID_CAP=n): a smaller persistent grid -- 256 = one workgroup per CU -- to see what an item costs
      // a workgroup that has its CU to itself (tools/dbg/strip_trace.py); the co-residency rule above still holds for the smaller grid
      static const long capg = [] { const char *e = getenv("MLPG_STRIP_GRID_CAP"); return e ? atol(e) : 0L; }();
      if (capg >= 16 * kMaxLists && capg < grid && R <= capg / (2 * kMaxLists)) grid = capg;
    }
    // The start ramp (round 6).  Left alone all 512 workgroups load their first item at once: the in-kernel trace
    // (tools/dbg/strip_trace.py, profiles/r06_strip_trace_*.txt) shows 512 of them loading for the first 12 us -- the memory system
    // overrun -- and then none at all for 5 us while every one of them runs its level-2/3 chain.  Spreading the FIRST items of a
    // list's workgroups over about one item period in ticket order (neighbouring strips then start a fraction of a microsecond
    // apart, which the wait for the right-hand neighbour absorbs) removes that first pile-up: config 2 under bench.py's protocol,
    // interleaved, nine pairs: 0.2137-0.2166 -> 0.2098-0.2115 ms per step (-2.1 %) at 16 / 20 / 24 us alike -- but where every strip
    // waits for its whole utterance (tight dynamic variances) a late start is only late: 0.257 -> 0.265 ms, and the float32 backward
    // loses 3 % as well, so the ramp ships switched off (MLPG_STRIP_STAGGER_DEFAULT_US).  It does NOT keep the
    // workgroups apart: a strip waits for its neighbours' records, so the workgroups that hold an utterance's strips finish together
    // and draw their next tickets together, and by the third item the loading count swings between 80 and 330 again (period 26-28
    // us, autocorrelation 0.6-0.8) as it does without the ramp.  Measured against that and not kept (profiles/r06_notes.md): a cap
    // on the workgroups of an XCD that may be in level 1 at a time (12-40 % slower: the waits add up along an utterance), the two
    // workgroups of a CU taking turns at level 1 (no gain), a chunk's last six frames by LDS-DMA (no gain: level 1 is not short of
    // bytes in flight).  Only launches that give every workgroup several items.  MLPG_STRIP_STAGGER_US overrides the span (0: off).
    a.stagger = 0;
    a.wpl = (int)(grid / a.nlists);
    {
      static const double us = [] { const char *e = getenv("MLPG_STRIP_STAGGER_US"); return e ? atof(e) : (double)MLPG_STRIP_STAGGER_DEFAULT_US; }();
      if (us > 0 && a.wpl >= 2 && nitems >= 3 * grid) a.stagger = (int)(us * 100.0);
    }
    note_launch(TR ? kCountStripTr : MULTI ? kCountStripMulti : kCountStrip);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kW * 64), kLdsBytes, st, p, ws, a);
    MLPG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((verdict_kernel<TIN, TOUT, BWD, MULTI, TR>), dim3((unsigned)((nsg + 3) / 4)), dim3(256), 0, st, p, ws, a);
    MLPG_HIP_CHECK(hipGetLastError());
    return 0;
  };
  if constexpr (MULTI) {
    // several streams side by side on the lanes: forward, per-frame variances, three windows (the caller checked)
    if constexpr (TR) {  // the transposed form also with global (D,) and unit variances
      if (p.var_mode == MLPG_HIP_VAR_GLOBAL) return go(strip_kernel<TIN, TOUT, false, MLPG_HIP_VAR_GLOBAL, true, true, true>);
      if (p.var_mode == MLPG_HIP_VAR_UNIT) return go(strip_kernel<TIN, TOUT, false, MLPG_HIP_VAR_UNIT, true, true, true>);
    }
    return go(strip_kernel<TIN, TOUT, false, MLPG_HIP_VAR_FRAME, true, true, TR>);
  } else {
    if (ws.nw == 3) {
      switch (p.var_mode) {
        case MLPG_HIP_VAR_FRAME: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_FRAME, false, true>);
        case MLPG_HIP_VAR_GLOBAL: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_GLOBAL, false, true>);
        default: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_UNIT, false, true>);
      }
    }
    switch (p.var_mode) {
      case MLPG_HIP_VAR_FRAME: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_FRAME, false, false>);
      case MLPG_HIP_VAR_GLOBAL: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_GLOBAL, false, false>);
      default: return go(strip_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_UNIT, false, false>);
    }
  }
}

template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, int R, int ndg, int dgw,
             bool zero_ctrl) {
  return launch_impl<TIN, TOUT, BWD, false>(st, p, ws, scratch_base, R, ndg, dgw, zero_ctrl, nullptr);
}
template <typename TIN, typename TOUT>
int launch_multi_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, int R, int ndg, int dgw,
                   bool zero_ctrl, const StreamMap &smap) {
  return launch_impl<TIN, TOUT, false, true>(st, p, ws, scratch_base, R, ndg, dgw, zero_ctrl, &smap);
}
template <typename TIN, typename TOUT>
int launch_tr_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, int R, bool zero_ctrl, const StreamMap &smap) {
  return launch_impl<TIN, TOUT, false, true, true>(st, p, ws, scratch_base, R, 1, 64, zero_ctrl, &smap);
}

}  // namespace strip
}  // namespace mlpg
