// Pipelined strip MLPG kernel (algo = MLPG_HIP_ALGO_PIPE; round 3).
//
// Same mathematics as mlpg_strip_impl.h (three-level substructured LDL^T: chunk of 16 frames in registers, strip
// separators through LDS, the utterance's strips through HBM records with a windowed level 3) -- what changes is WHO
// does what WHEN, so that the memory pipe never waits for the latency chain of levels 2 and 3:
//
//   workgroup = 4 wavefronts on the 4 SIMDs of one CU, one workgroup per CU, persistent;
//   wavefronts 0..2 are CHUNK wavefronts (lane = static dim, 16 frames each: a strip is 48 frames): level 1 (loads,
//     assembly, elimination: strip::assemble_eliminate) of item s, THEN the back-substitution and the stores of item
//     s-1 -- whose separator solutions arrived while item s was loading.  The factor of item s-1 stays in registers
//     meanwhile (one wavefront per SIMD: 512 registers per lane, the compiler parks what does not fit the 256
//     architectural ones in AGPRs);
//   wavefront 3 is the CHAIN wavefront: an event loop that, per item, runs level 2 over the three chunk records,
//     publishes the strip's record, watches the neighbours' flags, stages their records, runs the level-3 sweep and the
//     level-2 back-substitution, and hands the separator solutions to the chunk wavefronts through LDS.  It also draws
//     the tickets.  Nothing in a chunk wavefront's instruction stream waits for HBM round trips of the protocol.
//   No __syncthreads in the item loop: the wavefronts meet through monotonic sequence words in LDS.
//
// Inter-workgroup protocol: as the strip kernel's (agent-scope records, a flag per strip, an arrival counter per
// system group, control words zeroed by verdict_kernel).  Progress: a workgroup holds at most two unfinished tickets
// and publishes the record of everything it holds without waiting for anybody (the chain wavefront serves "records
// ready -> level 2 -> publish" also while it is polling for an older item); a third ticket is drawn only when the
// oldest item's neighbours have all arrived.  With G resident workgroups the lowest unfinished ticket of a list
// therefore always finds tickets up to itself + 2 G_list - 1 drawn and published: the launcher keeps the strips of
// one utterance (R) below that (see launch_t).
#pragma once
#include "mlpg_strip_impl.h"

#ifdef MLPG_PIPE_TIMING
#define PIPE_TICK(k)                                                  \
  do {                                                                \
    const long long t_now_ = (long long)__builtin_readcyclecounter(); \
    tq[k] += t_now_ - t_prev;                                         \
    t_prev = t_now_;                                                  \
  } while (0)
#else
#define PIPE_TICK(k) do {} while (0)
#endif

#ifdef MLPG_PIPE_TRACE
// realtime stamps (100 MHz, comparable across CUs) per item into the status array: workgroups 0 .. 31, 24 items, 10 stamps
#define PIPE_STAMP(item, k)                                                                                     \
  do {                                                                                                          \
    if (lane == 0 && p.status && blockIdx.x < 32 && (item) < 24)                                                \
      p.status[(blockIdx.x * 24 + (item)) * 10 + (k)] = (int)((long long)__builtin_amdgcn_s_memrealtime() & 0x3fffffff); \
  } while (0)
#else
#define PIPE_STAMP(item, k) do {} while (0)
#endif

namespace mlpg {
namespace pipe {

using strip::kM;
using strip::kN;
using strip::kRec;
using strip::kStage;
using strip::kFac;
using strip::kCtrlLine;
using strip::kMaxLists;
using strip::kSpinLimit;
using strip::kLocal;
using strip::kRouteTol;
using strip::kDampTol;
using strip::S2;
using strip::M2;
using strip::V2;
using strip::Args;
using strip::Window;
using strip::Order;
using namespace strip;  // the 2x2 block helpers, record slots, ld_agent / st_agent, fast_rcp

constexpr int kC = 3;                 // chunk wavefronts per workgroup = chunks per strip
constexpr int kStripFrames = kC * kM;  // 48
constexpr int kThreads = (kC + 1) * 64;

// ---- LDS ----
constexpr size_t kRecBytes1 = (size_t)kRec * 64 * 8;                   // one record: 7 KB
constexpr size_t oRec = 0;                                             // [2][kC][kRec][64]  level-1 records, by item parity
constexpr size_t oOwn = oRec + 2 * kC * kRecBytes1;                    // [2][kRec][64]      the strip's own level-2 record
constexpr size_t oFac = oOwn + 2 * kRecBytes1;                         // [2][kC-1][kFac][64]
constexpr size_t oU = oFac + 2 * (size_t)(kC - 1) * kFac * 64 * 8;     // [2][kC+1][2][64]   separator solutions
constexpr size_t oStage = oU + 2 * (size_t)(kC + 1) * 2 * 64 * 8;      // [kStage][kRec][64] level-3 staging (chain wavefront)
constexpr size_t oY = oStage + (size_t)kStage * kRecBytes1;            // [2][kC][kY=5][64]  right-hand sums, by item parity and consumer
constexpr size_t oCtl = oY + (2 * (size_t)kC + 2) * 5 * 64 * 8;        // ints   (+ a write-only slot and a slot of zeros)
constexpr size_t kLdsBytes = oCtl + 64 * sizeof(int);
static_assert(kLdsBytes <= 160 * 1024, "LDS");

// control words (ints) at oCtl
enum {
  cTk = 0,        // [4] ticket ring: item id (g * R + r) of sequence number s at [s & 3]; -1 = no more items
  cSeqTk = 4,     // tickets posted so far
  cSeqRec = 5,    // [kC] level-1 records written by chunk wavefront w
  cSeqU = 8,      // items whose separator solutions have been posted
  cTimedOut = 9,  // [2] by item parity: a wait of this item timed out
  cSeqY = 12,     // [kC] hand-overs written by chunk wavefront w (to wavefront w-1)
};

__device__ __forceinline__ int lds_ld(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS operations of one wavefront are executed in program order; this only keeps the COMPILER from moving data
// accesses across the flag access (and drains this wavefront's outstanding LDS operations)
__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct Item {  // everything derived from a ticket (wave-uniform)
  int g, r, b, dg, T, Ract, d0, nd;
  int pad, xwg;  // (ints: the struct is carried in scalar registers)
};
__device__ __forceinline__ Item decode(const Problem &p, const Args &a, int id) {
  Item it;
  it.g = id / a.R;
  it.r = id - it.g * a.R;
  it.b = it.g / a.ndg;
  it.dg = it.g - it.b * a.ndg;
  int T = p.lengths ? p.lengths[it.b] : p.Tmax;
  T = T < 0 ? 0 : (T > p.Tmax ? p.Tmax : T);
  it.T = T;
  it.Ract = (T + kStripFrames - 1) / kStripFrames;
  it.pad = it.r >= it.Ract;
  it.xwg = it.Ract > 1;
  it.d0 = it.dg * a.dgw;
  it.nd = p.sd - it.d0 < a.dgw ? p.sd - it.d0 : a.dgw;
  return it;
}


// ---- level 1 of the pipelined kernel ----------------------------------------------------------------
// One chunk wavefront owns the ROWS f0 .. f0+M-1 of its system (lane = static dim) and loads the FRAMES f0-1 .. f0+M-2:
// frame t feeds rows t-1, t, t+1, so the only contributions the wavefront cannot form itself are those of frames
// f0+M-1 and f0+M to its last two rows (its separator).  Those frames are the first two frames of the NEXT chunk
// wavefront of the strip, which hands the five sums over (Y: LDS, one sequence word per producer) as soon as it has
// seen them -- at the START of its stream, while the consumer needs them at the END of its own: nobody waits.  Only
// the strip's last chunk wavefront loads its two right-hand frames itself (they belong to another workgroup).  A
// strip of 48 frames is read as 50 frames; the strip kernel read 54 (18 per wavefront).
// The stream is uniform code for all wavefronts: a wavefront that gets its right-hand sums from LDS still issues the
// two loads (of its own last frame: a cache hit, no HBM traffic) and weighs them with 0.
//
// The stream is split into a PROLOGUE (the first kRing frames' loads: no arithmetic, needs nothing but the ticket)
// and the BODY, so that the prologue of item s+1 is in flight while the wavefront back-substitutes and stores item s-1.
#ifndef MLPG_PIPE_RING_F64
#define MLPG_PIPE_RING_F64 6
#endif
#ifndef MLPG_PIPE_RING_F32
#define MLPG_PIPE_RING_F32 8
#endif
template <typename TIN> struct PipeRing { static constexpr int value = MLPG_PIPE_RING_F64; };
template <> struct PipeRing<float> { static constexpr int value = MLPG_PIPE_RING_F32; };
constexpr int kY = 5;  // doubles per lane handed to the previous chunk wavefront: Pd[M-2], P1[M-2], rhs[M-2], Pd[M-1], rhs[M-1]

template <typename TIN>
struct L1Args {  // wave-uniform description of one chunk (everything the stream needs besides the ring)
  __amdgpu_buffer_rsrc_t mrs, vrs, grs;
  const TIN *vglob;
  unsigned loff, ldi_bytes, win_bytes, ldg_bytes;
  int f0, T, mw;
  int last;      // this wavefront loads its two right-hand frames itself (last chunk of the strip)
  int lo[3], hi[3], cl[3], ch[3];  // live frames [lo, hi) of each window, load clamp [cl, ch)
};
template <typename TIN>
__device__ __forceinline__ void l1_windows(L1Args<TIN> &A) {
  const int T = A.T, mw = A.mw;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    A.lo[w] = w ? mw : 0;
    A.hi[w] = w ? (mw != 0 && T - mw > mw ? T - mw : mw) : T;
    A.cl[w] = A.lo[w] < T ? A.lo[w] : T - 1;   // a window without live frames still loads (frame cl) and weighs 0
    A.ch[w] = A.hi[w] > A.cl[w] ? A.hi[w] : A.cl[w] + 1;
  }
}
// frame i (relative to f0) of window w -> the row it is loaded from
template <typename TIN, bool CLAMP>
__device__ __forceinline__ unsigned l1_soff(const L1Args<TIN> &A, const int i, const int w) {
  int t = A.f0 + i;
  if (i >= kM - 1) t = A.last ? t : A.f0 + kM - 2;  // right-hand frames of a wavefront that gets their sums from LDS
  if (CLAMP) t = t < A.cl[w] ? A.cl[w] : (t >= A.ch[w] ? A.ch[w] - 1 : t);
  return (unsigned)t * A.ldi_bytes + (unsigned)w * A.win_bytes;
}
template <typename TIN, bool BWD, int VM, bool CLAMP>
__device__ __forceinline__ void l1_load_frame(const L1Args<TIN> &A, TIN (&v)[3], TIN (&m)[3], const int i) {
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const unsigned soff = l1_soff<TIN, CLAMP>(A, i, w);
    if (VM == MLPG_HIP_VAR_FRAME) v[w] = ld_row<TIN>(A.vrs, soff, A.loff);
    if (!BWD) m[w] = ld_row<TIN>(A.mrs, soff, A.loff);
  }
}
template <typename TIN, bool BWD, int VM, int RING>
__device__ __forceinline__ void l1_prologue(const L1Args<TIN> &A, TIN (&rv)[RING][3], TIN (&rm)[RING][3]) {
#pragma unroll
  for (int sl = 0; sl < RING; ++sl) l1_load_frame<TIN, BWD, VM, true>(A, rv[sl], rm[sl], sl - 1);
}

// The body: accumulate frame by frame, refill the ring, eliminate row i as soon as frame i+1 is in (as
// strip::assemble_eliminate).  y_out / y_in: this lane's slots of the LDS hand-over (producer: predecessor's slot).
template <typename TIN, bool BWD, int VM, bool EDGE, int RING>
__device__ __forceinline__ bool l1_body(const L1Args<TIN> &A, TIN (&rv)[RING][3], TIN (&rm)[RING][3], const double (*wc)[9],
                                        const double one, double *y_out, int *y_out_seq, const double *y_in,
                                        const int *y_in_seq, const int seq, const int lane, double (&Pd)[kM], double (&P1)[kM],
                                        double (&P2)[kM], double (&rhs)[kM], double &ca, double &cb, double &cc,
                                        double (&rec)[kRec]) {
  const int f0 = A.f0, T = A.T;
  WinCoef k[3];
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    k[w] = win_coef<TIN, VM>(wc, w, A.vglob, (int)(A.win_bytes / sizeof(TIN)));
    if (VM == MLPG_HIP_VAR_UNIT) {  // the unit precision as an opaque per-lane value (see strip::assemble_eliminate)
      double t1 = one;
      asm volatile("" : "+v"(t1));
      k[w].tau_glob = t1;
    }
  }
  const double wR = A.last ? 1.0 : 0.0;  // weight of the two right-hand frames
  double y[kY];
  auto accumulate_frame = [&](const TIN (&v)[3], const TIN (&m)[3], const int i) __attribute__((always_inline)) {
    const int t = f0 + i;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      double tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(v[w]) : k[w].tau_glob;
      if (EDGE) tau *= (t >= A.lo[w] && t < A.hi[w]) ? 1.0 : 0.0;  // wave-uniform weight
      if (i >= kM - 1) tau *= wR;
      double tm = 0.0;
      if (!BWD) tm = tau * (double)m[w];
      const bool first = w == 0;  // first contribution to: Pd, rhs of row t+1; P1 of row t; P2 of row t-1
      if (i >= 0 && i < kM) {  // row t
        Pd[i] += k[w].c00 * tau;
        if (i < kM - 1) P1[i] = first ? k[w].cp0 * tau : P1[i] + k[w].cp0 * tau;
        if (!BWD) rhs[i] += k[w].c0 * tm;
      }
      if (i + 1 >= 0 && i + 1 < kM) {  // row t+1
        Pd[i + 1] = first ? k[w].cpp * tau : Pd[i + 1] + k[w].cpp * tau;
        if (!BWD) rhs[i + 1] = first ? k[w].cp * tm : rhs[i + 1] + k[w].cp * tm;
      }
      if (i - 1 >= 0 && i - 1 < kM) {  // row t-1
        Pd[i - 1] += k[w].cmm * tau;
        if (i - 1 < kM - 1) P1[i - 1] += k[w].c0m * tau;
        if (i - 1 < kM - 2) P2[i - 1] = first ? k[w].cpm * tau : P2[i - 1] + k[w].cpm * tau;
        if (!BWD) rhs[i - 1] += k[w].cm * tm;
      }
      // coupling of the chunk's first two rows to the previous chunk's separator:
      // ca = P[f0, f0-2], cb = P[f0, f0-1], cc = P[f0+1, f0-1]
      if (i == -1) {
        ca = first ? k[w].cpm * tau : ca + k[w].cpm * tau;
        cb = first ? k[w].cp0 * tau : cb + k[w].cp0 * tau;
        // ... and what this frame adds to the PREVIOUS chunk's rows M-2 (t-1) and M-1 (t)
        y[0] = first ? k[w].cmm * tau : y[0] + k[w].cmm * tau;
        y[1] = first ? k[w].c0m * tau : y[1] + k[w].c0m * tau;
        y[3] = first ? k[w].c00 * tau : y[3] + k[w].c00 * tau;
        if (!BWD) {
          y[2] = first ? k[w].cm * tm : y[2] + k[w].cm * tm;
          y[4] = first ? k[w].c0 * tm : y[4] + k[w].c0 * tm;
        }
      }
      if (i == 0) {
        cb += k[w].c0m * tau;
        cc = first ? k[w].cpm * tau : cc + k[w].cpm * tau;
        y[3] += k[w].cmm * tau;  // the previous chunk's row M-1 is this frame's row t-1
        if (!BWD) y[4] += k[w].cm * tm;
      }
    }
  };
  auto fix_row = [&](const int i) __attribute__((always_inline)) {
    const int f = f0 + i;
    const double live = f < T ? 1.0 : 0.0, live1 = f + 1 < T ? 1.0 : 0.0, live2 = f + 2 < T ? 1.0 : 0.0;
    Pd[i] = Pd[i] * live + (1.0 - live);
    if (i < kM - 1) P1[i] *= live1;
    if (i < kM - 2) P2[i] *= live2;
    rhs[i] *= live;
  };
  bool bad = false;
  double t00 = 0.0, t01 = 0.0, t11 = 0.0, h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
  auto elim_row = [&](const int i) __attribute__((always_inline)) {
    if (EDGE) {
      fix_row(i);
      if (i == 0) {
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

  static_assert(RING >= 2 && RING <= kM + 2, "ring depth");
  if (BWD) {
#pragma unroll
    for (int i = 0; i < kM; ++i) {
      int t = f0 + i;
      t = t >= T ? T - 1 : t;
      rhs[i] = (double)ld_row<TIN>(A.grs, (unsigned)t * A.ldg_bytes, A.loff);  // rows >= T are reset by fix_row
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // step S handles frame I = S - 1 in ring slot S % RING: accumulate, refill with frame I + RING, eliminate row I - 1
#define PIPE_STEP(S)                                                                                   \
  accumulate_frame(rv[(S) % RING], rm[(S) % RING], (S)-1);                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  if ((S) + RING < kM + 2) { l1_load_frame<TIN, BWD, VM, EDGE>(A, rv[(S) % RING], rm[(S) % RING], (S)-1 + RING); } \
  __builtin_amdgcn_sched_barrier(0);                                                                   \
  if ((S) == 1) {                                                                                      \
    /* frames f0-1 and f0 are in: hand their sums to the previous chunk wavefront (the strip's first */ \
    /* wavefront writes into a slot nobody reads: no branch in the stream) */                          \
    _Pragma("unroll")                                                                                  \
    for (int q = 0; q < kY; ++q) y_out[q * 64] = (BWD && (q == 2 || q == 4)) ? 0.0 : y[q];             \
    lds_order();                                                                                       \
    if (lane == 0) lds_st(y_out_seq, seq + 1);                                                         \
  }                                                                                                    \
  if ((S)-2 >= 0 && (S)-2 < kN) { elim_row((S)-2); }                                                   \
  __builtin_amdgcn_sched_barrier(0);
  PIPE_STEP(0) PIPE_STEP(1) PIPE_STEP(2) PIPE_STEP(3) PIPE_STEP(4) PIPE_STEP(5)
  PIPE_STEP(6) PIPE_STEP(7) PIPE_STEP(8) PIPE_STEP(9) PIPE_STEP(10) PIPE_STEP(11)
  PIPE_STEP(12) PIPE_STEP(13) PIPE_STEP(14) PIPE_STEP(15) PIPE_STEP(16) PIPE_STEP(17)
#undef PIPE_STEP
  static_assert(kM == 16, "the stream above is written out for 16-frame chunks");
  {
    // the right-hand sums come from the next chunk wavefront (it wrote them at the start of its own stream); the
    // strip's last wavefront formed them itself: it looks at its own sequence word and adds a slot of zeros
    while (lds_ld(y_in_seq) <= seq) __builtin_amdgcn_s_sleep(1);
    lds_order();
    Pd[kM - 2] += y_in[0 * 64];
    P1[kM - 2] += y_in[1 * 64];
    Pd[kM - 1] += y_in[3 * 64];
    if (!BWD) {
      rhs[kM - 2] += y_in[2 * 64];
      rhs[kM - 1] += y_in[4 * 64];
    }
  }
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


// ---- explicit parking in accumulation registers ------------------------------------------------------
// One wavefront per SIMD owns 512 registers per lane, of which only 256 (the architectural VGPRs) can be operands.
// The factor of the item that waits for its separators is parked in the other 256 (AGPRs) BY HAND: values of
// register class "a" that the allocator cannot mistake for something worth keeping in a VGPR.  (Left to itself
// -- plain double arrays live across the loop -- it shuffled 800 accvgpr moves per item and still spilled to scratch.)
struct Parked { unsigned lo, hi; };
__device__ __forceinline__ Parked park(const double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned l = (unsigned)u, h = (unsigned)(u >> 32);
  Parked r;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.lo) : "v"(l));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(r.hi) : "v"(h));
  return r;
}
__device__ __forceinline__ double unpark(const Parked &q) {
  unsigned l, h;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(q.lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(q.hi));
  return __longlong_as_double((long long)(((unsigned long long)h << 32) | l));
}
struct ParkedFactor {  // rows 0 .. kN-1 of 1/d, l1, l2, g and the three couplings
  Parked d[kN], l1[kN], l2[kN], g[kN], ca, cb, cc;
};

// back-substitution of one chunk straight from the parked factor (strip::backsub): x[0..kM) on return
__device__ __forceinline__ void backsub_parked(const ParkedFactor &F, double (&x)[kM], const V2 ul, const V2 u) {
  {
    const double ca = unpark(F.ca), cb = unpark(F.cb), cc = unpark(F.cc);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
    for (int i = 0; i < kN; ++i) {
      const double ba = (i == 0) ? ca : 0.0;
      const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
      const double va = ba - q1 * a1 - q3 * a2;
      const double vb = bb - q1 * b1 - q3 * b2;
      x[i] = unpark(F.g[i]) - (va * ul.x + vb * ul.y);
      a2 = a1; a1 = va;
      b2 = b1; b1 = vb;
      q3 = q2; q2 = unpark(F.l2[i]); q1 = unpark(F.l1[i]);
    }
  }
  double x1 = u.x, x2 = u.y;
#pragma unroll
  for (int i = kN - 1; i >= 0; --i) {
    const double xi = x[i] * unpark(F.d[i]) - unpark(F.l1[i]) * x1 - unpark(F.l2[i]) * x2;
    x[i] = xi;
    x2 = x1;
    x1 = xi;
  }
  x[kN] = u.x;
  x[kN + 1] = u.y;
}

// ---- the kernel ---------------------------------------------------------------------------------
template <typename TIN, typename TOUT, bool BWD, int VM>
__global__ __launch_bounds__(kThreads, 1) void pipe_kernel(Problem p, WinSet ws, Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *lds_rec = (double *)(smem + oRec);
  double *lds_own = (double *)(smem + oOwn);
  double *lds_fac = (double *)(smem + oFac);
  double *lds_u = (double *)(smem + oU);
  double *lds_stage = (double *)(smem + oStage);
  double *lds_y = (double *)(smem + oY);
  int *ctl = (int *)(smem + oCtl);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = a.R;
  const int sd = p.sd, Tmax = p.Tmax, mw = ws.mw, nw = ws.nw;
  const long ldi = p.ld_in, ldg = p.ld_gout, ldo = p.ld_out;

  if (tid < 64) ctl[tid] = 0;
  for (int q = tid; q < kY * 64; q += kThreads) lds_y[(size_t)(2 * kC + 1) * kY * 64 + q] = 0.0;  // the slot of zeros
  __syncthreads();

  if (wv < kC) {
    // =============================== chunk wavefront ===============================
    constexpr int kRing = PipeRing<TIN>::value;
    const unsigned ldi_bytes = (unsigned)ldi * (unsigned)sizeof(TIN), win_bytes = (unsigned)sd * (unsigned)sizeof(TIN);
    // wave-uniform description of a chunk of item `it` for this wavefront
    auto make_args = [&](const Item &it) __attribute__((always_inline)) {
      L1Args<TIN> A;
      const bool lane_ok = lane < it.nd;
      const int d = it.d0 + (lane_ok ? lane : it.nd - 1);  // idle lanes shadow the group's last dim (never stored)
      A.loff = (unsigned)(d - it.d0) * (unsigned)sizeof(TIN);
      A.mrs = make_rsrc(BWD ? (const TIN *)p.out : (const TIN *)p.mean + (size_t)it.b * Tmax * ldi + it.d0);
      A.vrs = make_rsrc(VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)it.b * Tmax * ldi + it.d0 : (const TIN *)p.out);
      A.grs = make_rsrc(BWD ? (const TIN *)p.grad_out + (size_t)it.b * Tmax * ldg + it.d0 : (const TIN *)p.out);
      A.vglob = VM == MLPG_HIP_VAR_GLOBAL ? (const TIN *)p.var + d : nullptr;
      A.ldi_bytes = ldi_bytes;
      A.win_bytes = win_bytes;
      A.ldg_bytes = (unsigned)ldg * (unsigned)sizeof(TIN);
      A.f0 = (it.r * kC + wv) * kM;
      A.T = it.T;
      A.mw = mw;
      A.last = wv == kC - 1;
      l1_windows(A);
      return A;
    };
    auto fetch_ticket = [&](const int q) __attribute__((always_inline)) {  // ticket number q -> item id (-1: no more items)
      while (lds_ld(ctl + cSeqTk) <= q) __builtin_amdgcn_s_sleep(2);
      lds_order();
      return __builtin_amdgcn_readfirstlane(lds_ld(ctl + cTk + (q & 3)));
    };

    ParkedFactor F;  // factor of the item awaiting its separators (in AGPRs)
    {
      const Parked z = park(0.0);
#pragma unroll
      for (int i = 0; i < kN; ++i) F.d[i] = F.l1[i] = F.l2[i] = F.g[i] = z;
      F.ca = F.cb = F.cc = z;
    }
    TIN rv[kRing][3], rm[kRing][3];  // the ring of frames in flight
    bool have_old = false;
    Item old_it = {};
#ifdef MLPG_PIPE_TIMING
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = (long long)__builtin_readcyclecounter();
    int n_items = 0;
#endif
    // the first item: ticket and prologue
    int id = fetch_ticket(0);
    Item cur = {};
    if (id >= 0) {
      cur = decode(p, a, id);
      if (!cur.pad) l1_prologue<TIN, BWD, VM, kRing>(make_args(cur), rv, rm);
    }
    for (int s = 0;; ++s) {
      PIPE_TICK(0);
      if (wv == 0 && id >= 0) PIPE_STAMP(s, 0);  // body start
      // ---- level 1 of item s (its first frames are already in flight) ----
      double Pd[kM], P1[kM], P2[kM], rhs[kM], ca = 0.0, cb = 0.0, cc = 0.0;
      int nid = -1;
      Item nxt = {};
      if (id >= 0) {
        const Item &it = cur;
        const bool lane_ok = lane < it.nd;
        if (it.pad) {
          // nothing but padding frames in this strip: zero-fill this chunk's rows (no record, no chain work)
          const int f0 = (it.r * kC + wv) * kM;
          const int d = it.d0 + (lane_ok ? lane : it.nd - 1);
          TOUT *out_b = (TOUT *)p.out + (size_t)it.b * Tmax * ldo;
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
          if (it.r == 0 && wv == 0 && lane_ok && p.status) p.status[(size_t)it.b * p.ld_status + d] = 0;  // T == 0
#pragma unroll
          for (int i = 0; i < kM; ++i) { Pd[i] = 1.0; P1[i] = P2[i] = rhs[i] = 0.0; }
        } else {
          const L1Args<TIN> A = make_args(it);
          double rec[kRec];
          // hand-over slots: this wavefront produces for wavefront wv-1 and consumes what wavefront wv+1 produced
          double *y_out = lds_y + ((size_t)(wv > 0 ? (s & 1) * kC + (wv - 1) : 2 * kC) * kY) * 64 + lane;
          const double *y_in = lds_y + ((size_t)(wv + 1 < kC ? (s & 1) * kC + wv : 2 * kC + 1) * kY) * 64 + lane;
          int *y_out_seq = ctl + cSeqY + wv;
          const int *y_in_seq = ctl + cSeqY + (wv + 1 < kC ? wv + 1 : wv);
          const bool interior = mw != 0 && A.f0 - 1 >= mw && A.f0 + kM < it.T - mw;
          bool bad;
          if (interior)
            bad = l1_body<TIN, BWD, VM, false, kRing>(A, rv, rm, a.wc, a.one, y_out, y_out_seq, y_in, y_in_seq, s, lane, Pd, P1,
                                                      P2, rhs, ca, cb, cc, rec);
          else
            bad = l1_body<TIN, BWD, VM, true, kRing>(A, rv, rm, a.wc, a.one, y_out, y_out_seq, y_in, y_in_seq, s, lane, Pd, P1,
                                                     P2, rhs, ca, cb, cc, rec);
          if (bad) rec[rD11] = __builtin_nan("");  // poisons every later level: the system is reported, not solved
          double *rp = lds_rec + ((size_t)((s & 1) * kC + wv) * kRec) * 64 + lane;
#pragma unroll
          for (int k = 0; k < kRec; ++k) rp[k * 64] = rec[k];
        }
        lds_order();
        if (lane == 0) lds_st(ctl + cSeqRec + wv, s + 1);
#ifdef MLPG_PIPE_TIMING
        ++n_items;
#endif
        PIPE_TICK(1);
        if (wv == 0) PIPE_STAMP(s, 1);  // body end, records written
        // ---- the next ticket, and its first frames into the ring (in flight during what follows) ----
        nid = fetch_ticket(s + 1);
        if (wv == 0) PIPE_STAMP(s, 2);  // next ticket known
        if (nid >= 0) {
          nxt = decode(p, a, nid);
          if (!nxt.pad) l1_prologue<TIN, BWD, VM, kRing>(make_args(nxt), rv, rm);
        }
        PIPE_TICK(4);
      }

      // ---- back-substitution and stores of item s-1 (one more turn of the loop after the last item) ----
      auto finish_item = [&](const Item &it, const int so) __attribute__((always_inline)) {
        while (lds_ld(ctl + cSeqU) <= so) __builtin_amdgcn_s_sleep(2);
        lds_order();
        PIPE_TICK(2);
        if (wv == 0) PIPE_STAMP(so, 3);  // separators arrived
        if (it.pad) return;
        const int f0 = (it.r * kC + wv) * kM;
        const bool lane_ok = lane < it.nd;
        const int d = it.d0 + (lane_ok ? lane : it.nd - 1);
        TOUT *out_b = (TOUT *)p.out + (size_t)it.b * Tmax * ldo;
        const double *up = lds_u + (size_t)(so & 1) * (kC + 1) * 2 * 64 + lane;
        const V2 ul = {up[(wv * 2) * 64], up[(wv * 2 + 1) * 64]};
        const V2 uo = {up[((wv + 1) * 2) * 64], up[((wv + 1) * 2 + 1) * 64]};
        const double sx = up[(kC * 2) * 64];
        const int timed_out = __builtin_amdgcn_readfirstlane(lds_ld(ctl + cTimedOut + (so & 1)));
        const bool sys_bad = !(sx == sx) || !(uo.x == uo.x) || !(ul.x == ul.x);  // NaN: some pivot of this system failed
        double Xr[kM];
        backsub_parked(F, Xr, ul, uo);
        // Verdict marks (see strip_kernel): a strip that met a failing pivot or a time-out marks its lanes in the
        // utterance's mask; verdict_kernel turns the marks into the reference's status and zero columns.
        if (wv == 0) {
          const unsigned long long m = timed_out ? ~0ull : __ballot(sys_bad && lane_ok);
          if (m != 0ull && lane == 0) {
            int *line = a.ctrl + (1 + kMaxLists + it.g) * kCtrlLine;
            __hip_atomic_fetch_or(line + 2, (int)(unsigned)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_or(line + 3, (int)(unsigned)(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (timed_out) __hip_atomic_store(line + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (it.r == 0 && lane_ok && p.status) p.status[(size_t)it.b * p.ld_status + d] = 0;
        }
        const bool zero_out = sys_bad || timed_out;
        if (!lane_ok) return;
        if (!BWD) {
#pragma unroll
          for (int i = 0; i < kM; ++i) {
            const int t = f0 + i;
            if (t < Tmax) out_b[(size_t)t * ldo + d] = (t < it.T && !zero_out) ? (TOUT)Xr[i] : (TOUT)0;
          }
        } else {
          // grad[t, w*sd+d] = tau_w[t] * (cm x[t-1] + c0 x[t] + cp x[t+1]): see strip_kernel's epilogue
          const unsigned loff = (unsigned)(d - it.d0) * (unsigned)sizeof(TIN);
          const __amdgpu_buffer_rsrc_t vrs = make_rsrc(
              VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)it.b * Tmax * ldi + it.d0 : (const TIN *)p.out);
          const TIN *vglob = VM == MLPG_HIP_VAR_GLOBAL ? (const TIN *)p.var + d : nullptr;
          const int T = it.T;
          auto load_w = [&](TIN (&v)[kM + 1], const int w) __attribute__((always_inline)) {
            if (VM != MLPG_HIP_VAR_FRAME) return;
#pragma unroll
            for (int i = -1; i < kM; ++i) {
              int t = f0 + i;
              t = t < 0 ? 0 : (t >= T ? T - 1 : t);
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
              if (i == -1 && f0 >= T) continue;
              if (i == kM - 1 && t != T - 1) continue;  // the next chunk writes it
              const bool lv = w ? (mw != 0 && t >= mw && t < T - mw) : true;
              double tau = 0.0;
              if (lv) tau = VM == MLPG_HIP_VAR_FRAME ? tau_of<TIN>(v[i + 1]) : tau_glob;
              const double xm = (i == -1) ? ul.x : ((i == 0) ? ul.y : Xr[i > 0 ? i - 1 : 0]);
              const double x0 = (i == -1) ? ul.y : Xr[i >= 0 ? i : 0];
              const double xp = (i == kM - 1) ? 0.0 : Xr[i + 1];
              const double gval = tau * (cm * xm + c0 * x0 + cp * xp);
              ow[(size_t)t * ldo] = zero_out ? (TOUT)0 : (TOUT)gval;
            }
          };
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
      };
      if (have_old) finish_item(old_it, s - 1);
      if (wv == 0 && have_old) PIPE_STAMP(s - 1, 4);  // back-substitution and stores issued
#ifdef MLPG_PIPE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      PIPE_TICK(3);
      if (id < 0) break;
#pragma unroll
      for (int i = 0; i < kN; ++i) { F.d[i] = park(Pd[i]); F.l1[i] = park(P1[i]); F.l2[i] = park(P2[i]); F.g[i] = park(rhs[i]); }
      F.ca = park(ca); F.cb = park(cb); F.cc = park(cc);
      have_old = true;
      old_it = cur;
      cur = nxt;
      id = nid;
    }
#ifdef MLPG_PIPE_TIMING
    // profiling build: mean cycles per item of (rotation, level-1 body, wait for the separators, back-substitution + stores,
    // next ticket + prologue)
    if (lane == 0 && p.status && blockIdx.x < 64) {
      for (int k = 0; k < 5; ++k) p.status[(blockIdx.x * 4 + wv) * 8 + k] = (int)(tq[k] / (n_items > 0 ? n_items : 1));
      p.status[(blockIdx.x * 4 + wv) * 8 + 7] = n_items;
    }
#endif
    return;
  }

  // =============================== chain wavefront ===============================
  const int xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;  // hwreg(HW_REG_XCC_ID, 0, 4)
  // ---- tickets: one list per XCD (or one list), own list first ----
  int lst_k = 0;            // lists tried so far
  bool exhausted = false;
  int n_post = 0;           // tickets posted (valid items)
  auto post_ticket = [&]() {
    if (exhausted) return;
    int id = -1;
    while (lst_k < a.nlists) {
      const int lst = (xcd + lst_k) % a.nlists;
      const int lim = ((a.nsg - lst + a.nlists - 1) / a.nlists) * R;  // items of this list
      int *ticket = a.ctrl + (1 + lst) * kCtrlLine;
      // one round trip (an exhausted list is over-drawn by at most one ticket per workgroup and list: harmless,
      // verdict_kernel re-zeroes the words)
      int tk = lim;
      if (lane == 0) tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tk = __builtin_amdgcn_readfirstlane(tk);
      if (tk < lim) {
        id = ((tk / R) * a.nlists + lst) * R + tk % R;
        break;
      }
      ++lst_k;
    }
    if (lane == 0) lds_st(ctl + cTk + (n_post & 3), id);
    lds_order();
    if (lane == 0) lds_st(ctl + cSeqTk, n_post + 1);
    if (id < 0) exhausted = true;
    else ++n_post;
  };

  // per-item chain state, by item parity (wave-uniform; the two items in flight have different parity)
  int st_state[2] = {0, 0}, st_route[2] = {0, 0}, st_spins[2] = {0, 0};
  Item st_it0 = {}, st_it1 = {};
  enum { sNone = 0, sWaitWindow, sWaitFull, sReady };  // sReady: separator solutions are in lds_u, only the in-order post is left

  int n_l2 = 0, n_fin = 0;
#ifdef MLPG_PIPE_TIMING
  long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_prev = (long long)__builtin_readcyclecounter();
#endif
  post_ticket();
  post_ticket();
  PIPE_TICK(0);

  // ---- level 2 of item n_l2 (its three level-1 records are in LDS), publish, route ----
  auto level2_publish = [&](const int q) __attribute__((always_inline)) {
    const int par = q & 1;
    const int id = __builtin_amdgcn_readfirstlane(lds_ld(ctl + cTk + (q & 3)));
    const Item it = decode(p, a, id);
    if (par) st_it1 = it;
    else st_it0 = it;
    st_spins[par] = 0;
    if (lane == 0) lds_st(ctl + cTimedOut + par, 0);
    if (it.pad) {
      st_state[par] = sReady;
      return;
    }
    const bool lane_ok = lane < it.nd;
    const double *recs = lds_rec + (size_t)par * kC * kRec * 64 + lane;
    auto R_ = [&](int j, int k) { return recs[(j * kRec + k) * 64]; };
    S2 E = {R_(0, rD11), R_(0, rD12), R_(0, rD22)};
    V2 gg = {R_(0, rF1), R_(0, rF2)};
    M2 V = {R_(0, rL11), R_(0, rL12), R_(0, rL21), R_(0, rL22)};
    if (it.r == 0) V = {0.0, 0.0, 0.0, 0.0};
    S2 Ts = {R_(0, rT00), R_(0, rT01), R_(0, rT11)};
    V2 hs = {R_(0, rH0), R_(0, rH1)};
    E = sub(E, S2{R_(1, rT00), R_(1, rT01), R_(1, rT11)});
    gg = sub(gg, V2{R_(1, rH0), R_(1, rH1)});
    bool bad2 = false;
    double *facs = lds_fac + (size_t)par * (kC - 1) * kFac * 64 + lane;
#pragma unroll
    for (int j = 0; j + 1 < kC; ++j) {
      const S2 Einv = sym_inv(E, bad2);
      const M2 L = {R_(j + 1, rL11), R_(j + 1, rL12), R_(j + 1, rL21), R_(j + 1, rL22)};
      const M2 Mn = mul_ms(L, Einv);
      const M2 EV = mul_sm(Einv, V);
      const V2 c = mul_sv(Einv, gg);
      Ts = add(Ts, mul_mtm_sym(V, EV));
      hs = add(hs, mul_mtv(V, c));
      double *f = facs + (size_t)j * kFac * 64;
      f[0 * 64] = c.x; f[1 * 64] = c.y;
      f[2 * 64] = EV.a; f[3 * 64] = EV.b; f[4 * 64] = EV.c; f[5 * 64] = EV.d;
      f[6 * 64] = Mn.a; f[7 * 64] = Mn.b; f[8 * 64] = Mn.c; f[9 * 64] = Mn.d;
      S2 Dn = {R_(j + 1, rD11), R_(j + 1, rD12), R_(j + 1, rD22)};
      V2 Fn = {R_(j + 1, rF1), R_(j + 1, rF2)};
      if (j + 2 < kC) {
        Dn = sub(Dn, S2{R_(j + 2, rT00), R_(j + 2, rT01), R_(j + 2, rT11)});
        Fn = sub(Fn, V2{R_(j + 2, rH0), R_(j + 2, rH1)});
      }
      E = sub(Dn, mul_mmt_sym(Mn, L));
      gg = sub(Fn, mul_mv(Mn, gg));
      V = neg(mul_mm(Mn, V));
    }
    if (bad2) E.a = __builtin_nan("");
    if (!it.xwg) {
      // the utterance is this one strip: solve its last separator here; the level-2 back-substitution follows at once
      bool bad3 = false;
      const S2 Ai = sym_inv(E, bad3);
      V2 sig = mul_sv(Ai, gg);
      if (bad3) sig.x = __builtin_nan("");
      double *up = lds_u + (size_t)par * (kC + 1) * 2 * 64 + lane;
      up[0] = 0.0; up[64] = 0.0;
      up[(kC * 2) * 64] = sig.x; up[(kC * 2 + 1) * 64] = sig.y;
      V2 un = sig;
      const V2 sprev = {0.0, 0.0};
#pragma unroll
      for (int j = kC - 2; j >= 0; --j) {
        const double *f = facs + (size_t)j * kFac * 64;
        const V2 c = {f[0 * 64], f[1 * 64]};
        const M2 EV = {f[2 * 64], f[3 * 64], f[4 * 64], f[5 * 64]};
        const M2 Mn = {f[6 * 64], f[7 * 64], f[8 * 64], f[9 * 64]};
        const V2 uj = sub(sub(c, mul_mv(EV, sprev)), mul_mtv(Mn, un));
        up[((j + 1) * 2) * 64] = uj.x; up[((j + 1) * 2 + 1) * 64] = uj.y;
        un = uj;
      }
      st_state[par] = sReady;
      return;
    }
    // publish the strip's record, then announce it
    double *rp = a.rec + ((size_t)it.g * R + it.r) * (kRec * 64) + lane;
    const double own[kRec] = {E.a, E.b, E.c, gg.x, gg.y, V.a, V.b, V.c, V.d, Ts.a, Ts.b, Ts.c, hs.x, hs.y};
#pragma unroll
    for (int k = 0; k < kRec; ++k) st_agent(rp + k * 64, own[k]);
    double *ownp = lds_own + (size_t)par * kRec * 64 + lane;
#pragma unroll
    for (int k = 0; k < kRec; ++k) ownp[k * 64] = own[k];
    // route, from this strip's own data alone (see strip_kernel)
    int route;
    {
      bool badr = false;
      const double t_own = 2.0 * amax4(mul_sm(sym_inv(E, badr), V));
      auto any_over = [&](const double tol) { return __ballot(lane_ok && !(t_own <= tol)) != 0ull; };  // NaN counts
      route = !any_over(kRouteTol) ? kLocal : !any_over(3e-6) ? 4 : !any_over(1.8e-3) ? 8 : !any_over(4.2e-2) ? 16 : 0;
      if (2 * route + 1 >= it.Ract) route = route > kLocal ? 0 : route;  // a window as wide as the utterance: sweep it all
    }
    st_route[par] = route;
    st_state[par] = route ? sWaitWindow : sWaitFull;
#ifndef MLPG_PIPE_FAKE_CHAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    int *cnt = a.ctrl + (1 + kMaxLists + it.g) * kCtrlLine;
    int *flags = a.ctrl + (1 + kMaxLists + a.nsg) * kCtrlLine + (size_t)it.g * flag_pitch(R);
    if (lane == 0) {
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flags + it.r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // ---- one look at what item (parity par) is waiting for ----
  auto arrived = [&](const int par) -> bool {
#ifdef MLPG_PIPE_FAKE_CHAIN  // timing experiment only: what if the protocol's round trips cost nothing
    return true;
#endif
    const Item it = par ? st_it1 : st_it0;
    int *cnt = a.ctrl + (1 + kMaxLists + it.g) * kCtrlLine;
    int *flags = a.ctrl + (1 + kMaxLists + a.nsg) * kCtrlLine + (size_t)it.g * flag_pitch(R);
    int f = 1;
    if (st_state[par] == sWaitFull) {
      f = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= it.Ract;
    } else {
      const Window w = local_window(it.r, it.Ract, st_route[par]);
      // windows wider than 64 strips cannot occur: route <= 16
      if (lane <= w.hiE - w.lo) f = __hip_atomic_load(flags + w.lo + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __ballot(f == 0) == 0ull;
  };

  // ---- level 3 + level-2 back-substitution of item (parity par), whose window (or utterance) has arrived.
  // Returns false if the window's result was rejected by the damping bound: the item then waits for the utterance.
  auto finish = [&](const int par, const int timed_out) -> bool {
    const Item it = par ? st_it1 : st_it0;
    const bool lane_ok = lane < it.nd;
    const int r = it.r, Ract = it.Ract, g = it.g;
    const bool full = st_state[par] == sWaitFull;
    const int route = st_route[par];
    const Window w = full ? Window{0, Ract - 1, 0} : local_window(r, Ract, route);
    const Order o = make_order(r, w.lo, w.hiE);
    const bool own_from_lds = !full && route <= kLocal;  // single-batch windows: the own record is in LDS already
    V2 sig = {0.0, 0.0}, sprev = {0.0, 0.0};
    double damp = 0.0;
    if (!timed_out) {
      // (the sweep of strip_kernel, with the staging done by this wavefront: the records of batch k+1 are loaded
      // into registers before batch k is consumed)
      S2 Ainv = {0.0, 0.0, 0.0};
      V2 av = {0.0, 0.0};
      M2 Mn = {0.0, 0.0, 0.0, 0.0};
      S2 Sb = {0.0, 0.0, 0.0}, Tn = {0.0, 0.0, 0.0};
      V2 sb = {0.0, 0.0}, hn = {0.0, 0.0};
      M2 Vn = {0.0, 0.0, 0.0, 0.0};
      double dt = w.lo > 0 ? 1.0 : 0.0, db = w.edge ? 1.0 : 0.0;
      bool bad3 = false;
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
      auto top_finalize = [&](const Rec &k) __attribute__((always_inline)) {
        const S2 A = sub(sub(Ej, k.T), mul_mmt_sym(Mn, Vj));
        const V2 aa = sub(sub(gj, k.h), mul_mv(Mn, av));
        Ainv = sym_inv(A, bad3);
        av = aa;
        Mn = mul_ms(k.V, Ainv);
        dt *= 2.0 * amax4(mul_sm(Ainv, Vj));
      };
      auto top_pend = [&](const Rec &k) __attribute__((always_inline)) { Ej = k.E; gj = k.g; Vj = k.V; };
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
      // Straight-line staging, kLoad records (56 loads: the hardware counts at most 63 outstanding) per batch: every slot
      // of a batch is loaded and stored, slots beyond the batch's end repeat its last record (a branch around a group
      // of loads makes the compiler drain the loads before it at the join: the records would arrive one round trip
      // after the other).  In the 5-strip window the strip's own record (the last position) rides along from LDS.
      constexpr int kLoad = 4;
      static_assert(kLoad % 2 == 0 && kLoad + 1 <= kStage, "pairs must not straddle batches; the staging area holds a batch + 1");
      double sv[kLoad][kRec];
      auto stage_load = [&](const int p0, const int kn) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kLoad; ++q) {
          const int pos = p0 + (q < kn ? q : kn - 1);
          const double *rp = a.rec + ((size_t)g * R + row_of(o, pos)) * (kRec * 64) + lane;
#if defined(MLPG_PIPE_FAKE_CHAIN) && MLPG_PIPE_FAKE_CHAIN == 1
          (void)rp;
#pragma unroll
          for (int k = 0; k < kRec; ++k) sv[q][k] = lds_own[((size_t)par * kRec + k) * 64 + lane];
#else
#pragma unroll
          for (int k = 0; k < kRec; ++k) sv[q][k] = ld_agent(rp + k * 64);
#endif
        }
      };
      auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kLoad; ++q) {
#pragma unroll
          for (int k = 0; k < kRec; ++k) lds_stage[(q * kRec + k) * 64 + lane] = sv[q][k];
        }
      };
      PIPE_TICK(2);
      stage_load(0, o.npos < kLoad ? o.npos : kLoad);
      PIPE_TICK(5);  // staging loads issued
      for (int p0 = 0; p0 < o.npos;) {
        int kn = o.npos - p0 < kLoad ? o.npos - p0 : kLoad;
        stage_store();
        if (own_from_lds && p0 + kn == o.npos - 1) {
#pragma unroll
          for (int k = 0; k < kRec; ++k) lds_stage[(kn * kRec + k) * 64 + lane] = lds_own[((size_t)par * kRec + k) * 64 + lane];
          kn += 1;
        }
#ifdef MLPG_PIPE_TIMING
        lds_order();
#endif
        PIPE_TICK(6);  // staging loads landed + LDS writes
        if (p0 + kn < o.npos) stage_load(p0 + kn, o.npos - p0 - kn < kLoad ? o.npos - p0 - kn : kLoad);
        lds_order();
        int q = 0;
        while (q < kn) {
          const int pos = p0 + q;
          if (pos < 2 * o.m) {
            const int idx = pos >> 1;
            const Rec kt = rd(q), kb = rd(q + 1);
            if (idx == 0) {
              top_pend(kt);
              if (w.edge) bot_edge(kb);
              else bot_row(kb);
            } else {
              top_finalize(kt);
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
              if (idx == 0 && w.edge) bot_edge(k);
              else bot_row(k);
            }
            q += 1;
          } else {
            const Rec k = rd(q);
            if (o.nt > 0) top_finalize(k);
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
              sprev = sub(mul_sv(Ainv, av), mul_mtv(Mn, sig));
              dt *= __builtin_fmax(1.0, 2.0 * amax4(mul_sm(Binv, k.V)));
              db *= __builtin_fmax(1.0, 2.0 * amax4(mul_smt(Ainv, k.V)));
            }
            if (bad3) sig.x = __builtin_nan("");
            q += 1;
          }
        }
        lds_order();  // the reads of this batch are done before the next batch overwrites the staging area
        p0 += kn;
      }
      PIPE_TICK(7);  // sweep
      damp = dt > db ? dt : db;
      if (!full) {
        const bool full_range = w.lo == 0 && !w.edge;
        const bool lane_fine = !lane_ok || full_range || (damp < kDampTol && sig.x == sig.x);
#ifndef MLPG_PIPE_FAKE_CHAIN
        if (__ballot(!lane_fine) != 0ull) return false;  // the whole utterance is needed
#else
        (void)lane_fine;
#endif
      }
    }
    // level-2 back-substitution -> the separator solutions of the strip
    double *up = lds_u + (size_t)par * (kC + 1) * 2 * 64 + lane;
    const double *facs = lds_fac + (size_t)par * (kC - 1) * kFac * 64 + lane;
    up[0] = sprev.x; up[64] = sprev.y;
    up[(kC * 2) * 64] = sig.x; up[(kC * 2 + 1) * 64] = sig.y;
    V2 un = sig;
#pragma unroll
    for (int j = kC - 2; j >= 0; --j) {
      const double *f = facs + (size_t)j * kFac * 64;
      const V2 c = {f[0 * 64], f[1 * 64]};
      const M2 EV = {f[2 * 64], f[3 * 64], f[4 * 64], f[5 * 64]};
      const M2 Mn = {f[6 * 64], f[7 * 64], f[8 * 64], f[9 * 64]};
      const V2 uj = sub(sub(c, mul_mv(EV, sprev)), mul_mtv(Mn, un));
      up[((j + 1) * 2) * 64] = uj.x; up[((j + 1) * 2 + 1) * 64] = uj.y;
      un = uj;
    }
    return true;
  };

  // ---- the event loop ----
  for (;;) {
    bool progress = false;
    PIPE_TICK(4);  // idle / loop overhead
    // (1) the oldest published item: have its neighbours arrived?
    if (n_fin < n_l2) {
      const int par = n_fin & 1;
      bool ready = st_state[par] == sReady;
      int timed_out = 0;
      if (!ready) {
        const bool arr_ = arrived(par);
        PIPE_TICK(1);  // polls
        if (arr_) {
          PIPE_STAMP(n_fin, 7);  // neighbours arrived
          ready = true;
        } else if (++st_spins[par] > kSpinLimit) {
          ready = true;
          timed_out = 1;
          if (lane == 0) atomicAdd(a.ctrl, 1);
        }
        if (ready) {
          if (timed_out && lane == 0) lds_st(ctl + cTimedOut + par, 1);
          if (!finish(par, timed_out)) {
            st_state[par] = sWaitFull;  // rejected window: wait for the whole utterance, sweep again
            st_spins[par] = 0;
            ready = false;
            progress = true;
          }
        }
      }
      if (ready) {
        // The oldest item is done (nothing it still needs comes from another workgroup): only now may a third ticket
        // be drawn.  Drawn any earlier -- say when the window's flags arrived, before the damping bound accepted the
        // window -- the new item could sit unpublished behind an item that waits for the whole utterance, the new
        // item's own strip included.
        PIPE_TICK(2);  // level 3 + level-2 back-substitution
        PIPE_STAMP(n_fin, 8);  // level 3 done
        post_ticket();
        lds_order();
        if (lane == 0) lds_st(ctl + cSeqU, n_fin + 1);
        PIPE_TICK(0);  // ticket
        PIPE_STAMP(n_fin, 9);  // ticket + separators posted
        st_state[par] = sNone;
        ++n_fin;
        progress = true;
      }
    }
    // (2) the next item's level-1 records: level 2, publish (never waits for anybody)
    if (n_l2 < n_post) {
      bool all = true;
#pragma unroll
      for (int w = 0; w < kC; ++w) all = all && lds_ld(ctl + cSeqRec + w) > n_l2;
      if (all && n_l2 - n_fin < 2) {
        lds_order();
        PIPE_TICK(4);
        PIPE_STAMP(n_l2, 5);  // records seen
        level2_publish(n_l2);
        PIPE_STAMP(n_l2, 6);  // published
        ++n_l2;
        progress = true;
        PIPE_TICK(3);  // level 2 + publish
      }
    }
    if (exhausted && n_fin == n_post) break;
    if (!progress) __builtin_amdgcn_s_sleep(MLPG_STRIP_POLL_SLEEP);
  }
#ifdef MLPG_PIPE_TIMING
  // (ticket, polls, level 3, level 2 + publish, idle) cycles per item
  if (lane == 0 && p.status && blockIdx.x < 64) {
    for (int k = 0; k < 5; ++k) p.status[(blockIdx.x * 4 + 3) * 8 + k] = (int)(tq[k] / (n_post > 0 ? n_post : 1));
    p.status[(blockIdx.x * 4 + 3) * 8 + 7] = n_post;
    for (int k = 5; k < 8; ++k) p.status[64 * 4 * 8 + blockIdx.x * 4 + (k - 5)] = (int)(tq[k] / (n_post > 0 ? n_post : 1));
  }
#endif
}

// ---- launcher ------------------------------------------------------------------------------------
template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws, void *scratch_base, int R, int ndg, int dgw,
             bool zero_ctrl) {
  Args a;
  const int nsg = p.B * ndg;
  a.ctrl = (int *)scratch_base;
  a.rec = (double *)((char *)scratch_base + strip::ctrl_bytes(nsg, R));
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
    int resident = 0;
    if (int rc = strip::resident_grid((const void *)kern, kThreads, kLdsBytes, &resident)) return rc;
    // Co-residency (header comment): every workgroup holds two tickets; one list per XCD only while an utterance's
    // strips fit into what an XCD's share of the grid holds, with a factor 2 to spare; one list otherwise, and the
    // whole grid must then hold an utterance (same margin) -- if not, the caller takes another kernel.
    a.nlists = (nitems >= 2L * resident && R <= resident / kMaxLists) ? kMaxLists : 1;
    if (nitems > resident && R > resident) return strip::kNotResident;
    if (zero_ctrl) MLPG_HIP_CHECK(hipMemsetAsync(a.ctrl, 0, strip::ctrl_ints(nsg, R) * sizeof(int), st));
    const long grid = nitems < resident ? nitems : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kThreads), kLdsBytes, st, p, ws, a);
    MLPG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((strip::verdict_kernel<TIN, TOUT, BWD>), dim3((unsigned)((nsg + 3) / 4)), dim3(256), 0, st, p, ws, a);
    MLPG_HIP_CHECK(hipGetLastError());
    return 0;
  };
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: return go(pipe_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_FRAME>);
    case MLPG_HIP_VAR_GLOBAL: return go(pipe_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_GLOBAL>);
    default: return go(pipe_kernel<TIN, TOUT, BWD, MLPG_HIP_VAR_UNIT>);
  }
}

}  // namespace pipe
}  // namespace mlpg
