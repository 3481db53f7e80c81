// fastdtw (Salvador & Chan 2007; semantics of slaypni/fastdtw's pure-Python
// implementation, restated in oracle/dtw_oracle.c) with the Euclidean local
// cost, one workgroup (8 wavefronts) per utterance pair.
//
// Replaces the per-pair fastdtw(x, y, radius, dist) call of
// DTWAligner.transform (reference: preprocessing/alignment.py:50).
//
// Per pair:
//   1. halving pyramid of both series in an HBM scratch (pairwise means, odd
//      tail dropped), all levels;
//   2. from the coarsest level (full DTW) down to level 0: per-row windows
//      [lo_i, hi_i] from the coarser path (interval form of __expand_window:
//      the union of (2r+1)^2 neighbourhoods along a monotone path is one
//      interval per row), then, in chunks of <= 63 rows:
//        a. the local cost of every window cell of chunk c+1 is computed by
//           the other wavefronts (rows read through L1/L2) WHILE wavefront 0 sweeps
//           chunk c: two cost buffers, one barrier per chunk;
//        b. the DP recurrence is swept along ANTI-DIAGONALS: lane r >= 1 owns row
//           i0 + r - 1 and at step s handles column s - r, so one step is exactly
//           one anti-diagonal.  Rows hand their values down with a DPP
//           wave_shr:1 register move -- no LDS round trip, no barrier inside
//           the sweep.  Lane 0 is the FEEDER: it replays the previous chunk's
//           last row (or the virtual row -1 of the level) through the same
//           instruction stream, so the first row of a chunk needs no special
//           case;
//        c. one back-pointer byte per window cell in LDS;
//   3. back-trace, parallel over SEGMENTS of 16 rows: every (segment, entry
//      column of its bottom row) candidate is traced by its own thread up to
//      the segment's top (which candidate of the segment above it reaches +
//      cell count); the candidates actually on the path are found by pointer
//      doubling over those links; one thread per segment re-traces its piece
//      and writes it at its final position.  ~100 dependent steps per level
//      instead of tx + ty.  (Windows too wide for the
//      candidate table fall back to a sequential wave-uniform walk.)  The new
//      path stays in LDS for the next level.
// Everything except the pyramid lives in LDS.
//
// Arithmetic is bit-compatible with the oracle: cost = sqrt(sum_k (x-y)^2)
// with separate multiply/add in ascending k; D = min(up+dt, left+dt, diag+dt)
// compared after the add, first minimum wins (up, left, diagonal).
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

#ifdef MLPG_DTW_TIMING
#define DTW_TICK(k)                                                    \
  do {                                                                 \
    const long long t_now_ = (long long)__builtin_readcyclecounter(); \
    tq[k] += t_now_ - t_prev;                                          \
    t_prev = t_now_;                                                   \
  } while (0)
#else
#define DTW_TICK(k) do {} while (0)
#endif

namespace mlpg {
namespace {

struct DtwParams {
  const double *X, *Y;
  const int32_t *lenx, *leny;
  int N, Tx, Ty, D, radius;
  int32_t *path_i, *path_j, *path_len;
  double *cost;
  int *cu_tickets;    // 2048 counters, one per CU (any initial value): see "which wavefront sweeps"
  double *pyr;        // N * pyr_stride doubles: x levels >= 1, then y levels >= 1
  size_t pyr_stride;  // (Tx + Ty) * D
  int cellcap;        // window cells per level
  int hwcap;          // back-pointer halfwords per level
  int chunkcap;       // doubles per cost buffer without the rows' frames: handed-over row + its slack + the chunk's cells
  int pcap_lds;       // path entries kept in LDS (levels >= 1)
  int tier;           // 0: one launch, capacities are bounds.  1: optimistic capacities, a pair that exceeds one is
                      // marked path_len = -1.  2: bounds again, only the marked pairs
  int dist_kind;      // MLPG_HIP_DIST_*
  double dist_scale;  // factor of MLPG_HIP_DIST_SCALED_L2_NP
};

constexpr int kMaxLevels = 20;
constexpr int kRows = 63;  // rows per chunk: lanes 1..63 of the sweeping wavefront (lane 0 feeds the row above)
constexpr int kSegMin = 4, kSegMax = 16;  // rows per back-trace segment: chosen per level (see the back-trace)
constexpr int kSlack = 8;        // +INF cells on both sides of a handed-over row (one block of 8 steps may overhang)
constexpr int kHandDummy = 72;   // 64 lanes x 8 steps of throw-away writes, overlapping
// threads per pair: wavefront 0 sweeps, all of them halve, compute local costs and back-trace.  512 for few pairs
// (latency), 256 for many (four pairs resident per CU: launch_fastdtw)

__device__ __forceinline__ double l2_cost(const double *__restrict__ a, const double *__restrict__ b, int D) {
  // The sum runs in ascending k with separate multiply and add (bit-compatible with the oracle); the
  // loads are issued eight at a time ahead of it so that the rows' memory latency is paid once per
  // block, not once per element.
  double acc = 0.0;
  int k = 0;
  for (; k + 8 <= D; k += 8) {
    double x[8], y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      x[q] = a[k + q];
      y[q] = b[k + q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double diff = __dsub_rn(x[q], y[q]);
      acc = __dadd_rn(acc, __dmul_rn(diff, diff));
    }
  }
  for (; k < D; ++k) {
    const double diff = __dsub_rn(a[k], b[k]);
    acc = __dadd_rn(acc, __dmul_rn(diff, diff));
  }
  return __dsqrt_rn(acc);
}

// Two neighbouring cells of one row at once: sqrt(sum_k (x_k - ya_k)^2) and sqrt(sum_k (x_k - yb_k)^2), each sum
// exactly as l2_cost (ascending k, separate multiply and add).  The rows are read 16 bytes at a time (they are 8-byte
// aligned: unaligned dwordx4 loads), 8 elements per row per batch, and the x row only once: a 25-dim pair of cells
// waits for memory twice, not eight times, and issues less than a third of the load instructions.  The costs under
// the sweep are bound by exactly that once four pairs share a CU.
typedef double dtw_d2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr int kCostPairs = 4;  // 16-byte loads per row per batch
__device__ __forceinline__ void l2_cost2(const double *__restrict__ x, const double *__restrict__ ya,
                                         const double *__restrict__ yb, int D, double *c0, double *c1) {
  double acc0 = 0.0, acc1 = 0.0;
  const int npair = D >> 1;
  // the odd last element is fetched first: it lands while the batches are worked on
  const double tx_ = x[D - 1], ta_ = ya[D - 1], tb_ = yb[D - 1];
  auto batch = [&](const dtw_d2 (&vx)[kCostPairs], const dtw_d2 (&vya)[kCostPairs], const dtw_d2 (&vyb)[kCostPairs], int cnt) {
#pragma unroll
    for (int q = 0; q < kCostPairs; ++q) {
      if (q < cnt) {
        double d0 = __dsub_rn(vx[q].x, vya[q].x), d1 = __dsub_rn(vx[q].x, vyb[q].x);
        acc0 = __dadd_rn(acc0, __dmul_rn(d0, d0));
        acc1 = __dadd_rn(acc1, __dmul_rn(d1, d1));
        d0 = __dsub_rn(vx[q].y, vya[q].y);
        d1 = __dsub_rn(vx[q].y, vyb[q].y);
        acc0 = __dadd_rn(acc0, __dmul_rn(d0, d0));
        acc1 = __dadd_rn(acc1, __dmul_rn(d1, d1));
      }
    }
  };
  int k = 0;
#pragma unroll 1
  for (; k + kCostPairs <= npair; k += kCostPairs) {  // full batches: one address per row, offsets in the instructions
    dtw_d2 vx[kCostPairs], vya[kCostPairs], vyb[kCostPairs];
    const dtw_d2 *px_ = (const dtw_d2 *)(x + 2 * k), *pa_ = (const dtw_d2 *)(ya + 2 * k), *pb_ = (const dtw_d2 *)(yb + 2 * k);
#pragma unroll
    for (int q = 0; q < kCostPairs; ++q) {
      vx[q] = px_[q];
      vya[q] = pa_[q];
      vyb[q] = pb_[q];
    }
    batch(vx, vya, vyb, kCostPairs);
  }
  if (k < npair) {  // the rest: indices past the end read the last pair again (not used)
    dtw_d2 vx[kCostPairs], vya[kCostPairs], vyb[kCostPairs];
#pragma unroll
    for (int q = 0; q < kCostPairs; ++q) {
      const int e = 2 * (k + q < npair ? k + q : npair - 1);
      vx[q] = *(const dtw_d2 *)(x + e);
      vya[q] = *(const dtw_d2 *)(ya + e);
      vyb[q] = *(const dtw_d2 *)(yb + e);
    }
    batch(vx, vya, vyb, npair - k);
  }
  if (D & 1) {
    const double d0 = __dsub_rn(tx_, ta_), d1 = __dsub_rn(tx_, tb_);
    acc0 = __dadd_rn(acc0, __dmul_rn(d0, d0));
    acc1 = __dadd_rn(acc1, __dmul_rn(d1, d1));
  }
  *c0 = __dsqrt_rn(acc0);
  *c1 = __dsqrt_rn(acc1);
}

// scale * sqrt(sum_k (a_k - b_k)^2) with the sum in numpy's order for a contiguous float64 vector of D <= 128
// elements (numpy/core/src/umath/loops_utils.h.src, pairwise sum: sequential below 8 elements, else eight
// partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail): what
// `scale * math.sqrt(((x - y) * (x - y)).sum(-1))` evaluates to, i.e. the reference's metrics.melcd(x, y) for two
// frames (metrics/__init__.py:27-59) with scale = 10 / ln 10 * sqrt 2.
template <int F>
__device__ __forceinline__ double np_term(double diff) {
  return F == 1 ? __builtin_fabs(diff) : __dmul_rn(diff, diff);
}
// F = 0: scale * sqrt(sum (a-b)^2)   1: scale * sum |a-b|   2: scale * sum (a-b)^2
template <int F>
__device__ __forceinline__ double np_cost(const double *__restrict__ a, const double *__restrict__ b, int D, double scale) {
  double res;
  if (D < 8) {
    res = 0.0;
    for (int k = 0; k < D; ++k) res = __dadd_rn(res, np_term<F>(__dsub_rn(a[k], b[k])));
  } else {
    double r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = np_term<F>(__dsub_rn(a[q], b[q]));
    int k = 8;
    for (; k < D - (D % 8); k += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = __dadd_rn(r[q], np_term<F>(__dsub_rn(a[k + q], b[k + q])));
    }
    res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])), __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; k < D; ++k) res = __dadd_rn(res, np_term<F>(__dsub_rn(a[k], b[k])));
  }
  return __dmul_rn(scale, F == 0 ? __dsqrt_rn(res) : res);
}
__device__ __forceinline__ double np_l2_cost(const double *__restrict__ a, const double *__restrict__ b, int D, double scale) {
  return np_cost<0>(a, b, D, scale);
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  *total = __shfl(incl, 63);
  return incl - v;
}

// lane r receives lane r-1's value, lane 0 receives +0.0 (bound_ctrl): one DPP move per 32-bit half
__device__ __forceinline__ double wave_shr1z(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int clamp_idx(int c, int hi) {  // median(c, -1, hi), hi >= 0
  int r;
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(c), "v"(hi));
  return r;
}

// Back-pointer codes are packed two bits per step of the sweep (see the sweep below).  rinfo[row] >> 32 is the row's
// cell-index base: cell (row, j) lives at bit pair (base + j - lo) of the halfword stream, most significant pair first
// within a halfword.
// TIE = MLPG_HIP_TIE_FIRST_MIN (upstream's pure-Python __dtw: the FIRST minimum of up, left, diagonal): the upper bit says
// "the diagonal beats left", the lower one "the better of the two beats up".
// TIE = MLPG_HIP_TIE_DIAG_LAST (the strict-less chain recalled for upstream's compiled _fastdtw: up only if it beats both
// others, else left only if it beats the diagonal, else the diagonal): upper bit "left beats the diagonal", lower bit
// "up beats the better of the two".
template <int TIE>
__device__ __forceinline__ unsigned bp_code(const unsigned short *__restrict__ bp16, int cidx) {
  const unsigned hw = bp16[cidx >> 3];
  const unsigned pair = (hw >> (14 - 2 * (cidx & 7))) & 3u;
  if (TIE == MLPG_HIP_TIE_FIRST_MIN) return (pair & 1u) ? 1u + (pair >> 1) : 0u;  // 0 up, 1 left, 2 diagonal
  return (pair & 1u) ? 0u : 2u - (pair >> 1);
}

// One back-trace walk from cell (bi, bj) up to (excluding) row `top`: follows the back-pointer
// codes, WRITE: stores the visited cells at positions wpos-1, wpos-2, ...  Returns the column
// reached in row top-1 (-2 if the walk leaves the window); *ncells = cells visited.
template <bool WRITE, int TIE, typename PT>
__device__ __forceinline__ int dtw_walk(const unsigned long long *__restrict__ rinfo, const unsigned short *__restrict__ bp16,
                                        PT *__restrict__ pth_i, PT *__restrict__ pth_j, int bi,
                                        int bj, int top, int wpos, int *ncells) {
  unsigned long long ri = rinfo[bi];
  unsigned long long rnext = bi > 0 ? rinfo[bi - 1] : 0ull;  // the row above is fetched one row ahead
  int rl = (int)(ri & 0xffffu);
  int cb = (int)(ri >> 32) - rl;  // cb + bj = cell index of (bi, bj)
  int cnt = 0;
  while (true) {
    ++cnt;
    const unsigned code = bp_code<TIE>(bp16, cb + bj);
    if (WRITE) {
      --wpos;
      pth_i[wpos] = (PT)bi;
      pth_j[wpos] = (PT)bj;
    }
    if (code != 0u) bj -= 1;  // left or diagonal
    if (code == 1u) {         // left: stay in the row
      if (bj < rl) { bj = -2; break; }
      continue;
    }
    bi -= 1;
    if (bi < top) break;
    ri = rnext;
    rnext = bi > 0 ? rinfo[bi - 1] : 0ull;
    rl = (int)(ri & 0xffffu);
    cb = (int)(ri >> 32) - rl;
    if (bj < rl || bj > (int)((ri >> 16) & 0xffffu)) { bj = -2; break; }
  }
  *ncells = cnt;
  return bj;
}

// (acc << 1) | (a < b): the compare lands in VCC and is shifted in by an add-with-carry
__device__ __forceinline__ unsigned shift_in_lt(unsigned acc, double a, double b) {
  unsigned r;
  asm("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
  return r;
}
// IEEE minNum as one instruction (the sweep's values are never NaN unless the inputs are)
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int kThreads, int TIE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void fastdtw_kernel(DtwParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x;
  const int Tx = p.Tx, Ty = p.Ty, D = p.D, r = p.radius;
  const int pcap = Tx + Ty, pcapL = p.pcap_lds;

  if (p.tier == 2 && p.path_len[n] != -1) return;  // retry launch: only the pairs the optimistic launch gave up on
  const int tx = p.lenx[n], ty = p.leny[n];
  int32_t *out_i = p.path_i + (size_t)n * pcap;
  int32_t *out_j = p.path_j + (size_t)n * pcap;
  if (tx < 1 || ty < 1 || tx > Tx || ty > Ty) {
    if (threadIdx.x == 0) {
      p.path_len[n] = 0;
      p.cost[n] = NAN;
    }
    return;
  }

  // ---- LDS carve (doubles / 64-bit words, ints, shorts) ----
  // Two cost buffers.  Buffer of chunk c: the row handed over by chunk c-1 (kSlack cells, its window, kSlack
  // cells), then the local costs of the chunk's window cells, every row framed by two +INF cells per side (costs are
  // read in pairs; a lane outside its window computes +INF without any select).
  const int dstride = p.chunkcap + 4 * kRows + 4;
  double *dchunk = (double *)smem;
  double *ddummy = dchunk + 2 * dstride;  // throw-away slots of the hand-over stores (lane l writes slots l .. l + 7)
  // per row: lo | hi << 16 | B << 32; B = offset of the row's first window cell (prefix sum of the widths) until the
  // row's chunk is swept, then the row's cell-index base in the back-pointer codes (see bp_code); entry ltx: the total
  unsigned long long *rinfo = (unsigned long long *)(ddummy + kHandDummy);
  unsigned *rw = (unsigned *)rinfo;
  auto LO = [&](int i) { return (int)(rw[2 * i] & 0xffffu); };
  auto HI = [&](int i) { return (int)(rw[2 * i] >> 16); };
  auto OFF = [&](int i) { return (int)rw[2 * i + 1]; };
  int *lvl_x = (int *)(rinfo + Tx + 1);
  int *lvl_y = lvl_x + kMaxLevels;
  int *bcast = lvl_y + kMaxLevels;  // [16]
  const int segcap = Tx / kSegMin + 3;
  int *segoff = bcast + 16;          // candidate-table offset of each back-trace segment
  int *segent = segoff + segcap;    // entry column (relative to the bottom row's window) chosen by the stitch
  int *segend = segent + segcap;    // end (exclusive) of the segment's piece in the path arrays
  unsigned short *cfirst = (unsigned short *)(segend + segcap);
  unsigned short *clast = cfirst + (Tx / 2 + 2);
  unsigned short *pth_i = clast + (Tx / 2 + 2);  // the path of a level >= 1 (level 0 goes straight to the output)
  unsigned short *pth_j = pth_i + pcapL;
  // back-pointer codes: 2 bits per sweep step, a halfword per 8 steps; per row the halfwords its window touches
  // (<= ((width + 6) >> 3) + 1; handed out chunk by chunk by the sweep), then one throw-away halfword per lane
  unsigned short *bp16 = (unsigned short *)(pth_j + pcapL);
  const int bp_dummy = p.hwcap;

  const double *x0 = p.X + (size_t)n * Tx * D;
  const double *y0 = p.Y + (size_t)n * Ty * D;
  double *px = p.pyr + (size_t)n * p.pyr_stride;
  double *py = px + (size_t)Tx * D;

#ifdef MLPG_DTW_TIMING
  long long tq[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // pyramid, windows+offsets, staging, costs, sweep, backtrace, output
  long long t_prev = (long long)__builtin_readcyclecounter();
#endif
  // number of halvings: level K is the first with a side < radius + 2 (full DTW there)
  int K = 0;
  while (K < kMaxLevels - 1 && (tx >> K) >= r + 2 && (ty >> K) >= r + 2) ++K;

  // ---- 0. which wavefront sweeps ----
  // The sweep is one wavefront issuing dependent instructions for most of the kernel; with several pairs resident on
  // a CU the sweepers must not pile up on one SIMD (wavefront w of every workgroup tends to land on the same one).
  // Every workgroup draws a ticket from a counter of its CU and lets the wavefront that runs on SIMD ticket % 4 sweep;
  // that wavefront becomes "wavefront 0" of everything below (thread ids are renumbered).
  {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // hwreg(HW_REG_HW_ID): simd [5:4], cu/sh/se [15:8]
    if (lane == 0) bcast[8 + (threadIdx.x >> 6)] = (int)((hwid >> 4) & 3u);
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;  // hwreg(HW_REG_XCC_ID, 0, 4)
      bcast[6] = p.cu_tickets ? (atomicAdd(&p.cu_tickets[(xcc << 8) | ((hwid >> 8) & 0xffu)], 1) & 3) : 0;
    }
  }
  __syncthreads();
  int sweeper = 0;
  for (int w = kThreads / 64 - 1; w >= 0; --w)
    if (bcast[8 + w] == bcast[6]) sweeper = w;
  const int hw_wave = threadIdx.x >> 6;
  const int tid = (hw_wave == sweeper ? 0 : (hw_wave == 0 ? sweeper : hw_wave)) * 64 + lane;
  const bool w0 = tid < 64;  // the sweeping wavefront

  // ---- 1. pyramid (levels 1..K) ----
  if (tid == 0) {
    int xo = 0, yo = 0;
    for (int k = 1; k <= K; ++k) {
      lvl_x[k] = xo;
      lvl_y[k] = yo;
      xo += (tx >> k) * D;
      yo += (ty >> k) * D;
    }
  }
  __syncthreads();
  {
    // up to four levels per round (one barrier each): an item reads a column of 16 / 8 / 4 / 2 consecutive source rows
    // and produces the 8 + 4 + 2 + 1 means above them -- every mean the same two-operand expression as level by level
    constexpr int kPyr = 4, kSpan = 1 << kPyr;
    const double *srcx = x0, *srcy = y0;
    int cx = tx, cy = ty;  // rows of the source level
    for (int k0 = 0; k0 < K;) {
      const int nl = K - k0 < kPyr ? K - k0 : kPyr, span = 1 << nl, half = span >> 1;
      const int nbx = ((cx >> 1) + half - 1) / half, nby = ((cy >> 1) + half - 1) / half;
      const int itx = nbx * D, items = (nbx + nby) * D;
      for (int e = tid; e < items; e += kThreads) {
        const bool isx = e < itx;
        const int ee = isx ? e : e - itx;
        const int blk = ee / D, c = ee - blk * D;
        const double *src = isx ? srcx : srcy;
        const int cnt0 = isx ? cx : cy;
        double *base_ = isx ? px : py;
        const int *lvl = isx ? lvl_x : lvl_y;
        double v[kSpan];
#pragma unroll
        for (int q = 0; q < kSpan; ++q) {
          const int row = blk * span + q;
          v[q] = (q < span && row < cnt0) ? src[(size_t)row * D + c] : 0.0;
        }
        // level k0 + j: rows blk * (span >> j) + q, q < span >> j, means of pairs of the level below (in place in v)
#pragma unroll
        for (int j = 1; j <= kPyr; ++j) {
          if (j <= nl) {
            double *dj = base_ + lvl[k0 + j];
            const int per = span >> j, cntj = cnt0 >> j;
#pragma unroll
            for (int q = 0; q < (kSpan >> j); ++q) {
              v[q] = __dadd_rn(v[2 * q], v[2 * q + 1]) * 0.5;
              const int rj = blk * per + q;
              if (q < per && rj < cntj) dj[(size_t)rj * D + c] = v[q];
            }
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      k0 += nl;
      srcx = px + lvl_x[k0];
      srcy = py + lvl_y[k0];
      cx >>= nl;
      cy >>= nl;
    }
  }

  DTW_TICK(0);
  int pstart = pcapL, pn = 0;  // current path = pth[pstart .. pstart+pn)
  double level_cost = INFINITY;
  bool fail = false;

  for (int k = K; k >= 0 && !fail; --k) {
    const int ltx = tx >> k, lty = ty >> k;
    const double *xk = k ? px + lvl_x[k] : x0;
    const double *yk = k ? py + lvl_y[k] : y0;

    // ---- 2a. per-row windows ----
    if (k == K) {
      for (int i = tid; i < ltx; i += kThreads) rw[2 * i] = (unsigned)(lty - 1) << 16;
    } else {
      const int cx = tx >> (k + 1);
      for (int q = tid; q < pn; q += kThreads) {
        const int pi = pth_i[pstart + q], pj = pth_j[pstart + q];
        if (q == 0 || pth_i[pstart + q - 1] != pi) cfirst[pi] = (unsigned short)pj;
        if (q == pn - 1 || pth_i[pstart + q + 1] != pi) clast[pi] = (unsigned short)pj;
      }
      __syncthreads();
      for (int i = tid; i < ltx; i += kThreads) {
        const int ci = i >> 1;
        const int r0 = ci - r < 0 ? 0 : ci - r;
        const int r1 = ci + r > cx - 1 ? cx - 1 : ci + r;
        const int a = 2 * ((int)cfirst[r0] - r);
        const int b = 2 * ((int)clast[r1] + r) + 1;
        rw[2 * i] = (unsigned)(a < 0 ? 0 : a) | ((unsigned)(b > lty - 1 ? lty - 1 : b) << 16);
      }
    }
    __syncthreads();

    // ---- 2b. row offsets (prefix sum of the widths) and the level's capacity checks; wavefront 0 ----
    if (w0) {
      const int rpl = (ltx + 63) / 64;
      const int b0 = lane * rpl < ltx ? lane * rpl : ltx;
      const int b1 = b0 + rpl < ltx ? b0 + rpl : ltx;
      int sum = 0;
      for (int i = b0; i < b1; ++i) sum += HI(i) - LO(i) + 1;
      int total;
      int run = wave_excl_scan(sum, lane, &total);
      for (int i = b0; i < b1; ++i) {
        rw[2 * i + 1] = (unsigned)run;
        run += HI(i) - LO(i) + 1;
      }
      if (lane == 0) {
        rw[2 * ltx] = 0u;
        rw[2 * ltx + 1] = (unsigned)total;
        bcast[3] = total > p.cellcap ? 1 : 0;  // (one flag word per check: [3] cells, [4] chunk table, [7] back-pointer
        bcast[7] = 0;                           //  words -- a reused word could be rewritten by wavefront 0 before a slow
                                                //  wavefront has read the previous verdict after its barrier)
        // virtual row -1 of the level: D[-1][-1] = 0, nothing else -- the row "handed over" to chunk 0
        dchunk[kSlack - 2] = INFINITY;
        dchunk[kSlack - 1] = INFINITY;
        dchunk[kSlack] = 0.0;
        dchunk[kSlack + 1] = INFINITY;
        dchunk[kSlack + 2] = INFINITY;
      }
    }
    __syncthreads();
    if (bcast[3]) fail = true;
    if (fail) break;
    const int ncell_lvl = OFF(ltx);

    DTW_TICK(1);
    // ---- 2c. chunk table (wavefront 0): chunk c = rows [cstart[c], cstart[c+1]), at most 63 rows; its cells and
    // the row handed over by the chunk before must fit a cost buffer.  The table lives where the coarse-path scratch
    // was (free until the next level) ----
    unsigned short *cstart = cfirst;
    // cells in front of a chunk's rows in its cost buffer: the handed-over row (the virtual row -1 has one cell)
    auto feed_cells = [&](int i0) { return (i0 > 0 ? HI(i0 - 1) - LO(i0 - 1) + 1 : 1) + 2 * kSlack; };
    // local costs of every window cell of the chunk of rows [i0, i0 + R) into its buffer, by threads [t0, t0 + nthr)
    auto chunk_costs_rows = [&](int i0, int R, double *buf, int t0, int nthr) {
      const int base = OFF(i0);
      const int ncell = OFF(i0 + R) - base;
      double *dst = buf + feed_cells(i0);
      if (p.dist_kind == MLPG_HIP_DIST_L2 && D >= 2) {
        // two cells per thread and round: cells 2t, 2t + 1 of the chunk.  Windows start on even columns and have even
        // widths except where they are clipped at the level's last column, so the two are neighbours in one row almost
        // always (one evaluation with the x row shared); a pair that a row boundary splits is evaluated cell by cell.
        for (int cc = 2 * (tid - t0); cc < ncell; cc += 2 * nthr) {
          int a = 0, b = R;
          while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (OFF(i0 + mid) - base <= cc) a = mid; else b = mid;
          }
          const int row = i0 + a, j = LO(row) + cc - (OFF(row) - base);
          const double *xr = xk + (size_t)row * D, *yr = yk + (size_t)j * D;
          const bool two = cc + 1 < ncell;
          if (!two || cc + 1 < OFF(row + 1) - base) {
            double c0, c1;
            l2_cost2(xr, yr, two ? yr + D : yr, D, &c0, &c1);
            dst[cc + 4 * a + 2] = c0;
            if (two) dst[cc + 4 * a + 3] = c1;
          } else {
            dst[cc + 4 * a + 2] = l2_cost(xr, yr, D);
            dst[cc + 1 + 4 * (a + 1) + 2] = l2_cost(xr + D, yk + (size_t)LO(row + 1) * D, D);
          }
        }
      } else
      for (int cc = tid - t0; cc < ncell; cc += nthr) {
        int a = 0, b = R;
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (OFF(i0 + mid) - base <= cc) a = mid; else b = mid;
        }
        const int row = i0 + a;
        const int j = LO(row) + cc - (OFF(row) - base);
        const double *xr = xk + (size_t)row * D, *yr = yk + (size_t)j * D;
        dst[cc + 4 * a + 2] = p.dist_kind == MLPG_HIP_DIST_L2             ? l2_cost(xr, yr, D)
                              : p.dist_kind == MLPG_HIP_DIST_SCALED_L2_NP ? np_cost<0>(xr, yr, D, p.dist_scale)
                              : p.dist_kind == MLPG_HIP_DIST_SCALED_L1_NP ? np_cost<1>(xr, yr, D, p.dist_scale)
                                                                          : np_cost<2>(xr, yr, D, p.dist_scale);
      }
      for (int a = tid - t0; a < R; a += nthr) {  // the +INF frame of every row
        const int fl = OFF(i0 + a) - base + 4 * a, fr = OFF(i0 + a + 1) - base + 4 * a + 2;
        dst[fl] = INFINITY;
        dst[fl + 1] = INFINITY;
        dst[fr] = INFINITY;
        dst[fr + 1] = INFINITY;
      }
    };
    auto chunk_costs = [&](int c, double *buf, int t0, int nthr) {
      chunk_costs_rows((int)cstart[c], (int)cstart[c + 1] - (int)cstart[c], buf, t0, nthr);
    };
    if (w0) {
      int i0 = 0, nc = 0, bad = 0;
      while (i0 < ltx) {
        const int base = OFF(i0), room = p.chunkcap - feed_cells(i0);
        // short first chunks: the pipeline (costs of chunk c+1 behind the sweep of chunk c) starts sooner
        const int rowcap = nc == 0 ? 8 : (nc == 1 ? 24 : kRows);
        const bool fits = (lane < rowcap) && (i0 + lane < ltx) && (OFF(i0 + lane + 1) - base <= room);
        const unsigned long long m = __ballot(fits);
        const int R = __ffsll((long long)~m) - 1;  // bit 63 is never set: R <= 63
        if (R < 1) { bad = 1; break; }
        if (lane == 0) cstart[nc] = (unsigned short)i0;
        ++nc;
        i0 += R;
      }
      if (lane == 0) {
        cstart[nc] = (unsigned short)ltx;
        bcast[2] = nc;
        bcast[4] = bad;
      }
    } else {
      // meanwhile the other wavefronts compute the local costs of chunk 0 -- its rows follow from the same rule (at
      // most 8 rows, cells within the buffer), which every thread evaluates for itself
      const int room0 = p.chunkcap - feed_cells(0);
      int R0 = 0;
      for (int r_ = 1; r_ <= 8 && r_ <= ltx && OFF(r_) <= room0; ++r_) R0 = r_;
      if (R0 >= 1) chunk_costs_rows(0, R0, dchunk, 64, kThreads - 64);
    }
    __syncthreads();
    if (bcast[4]) fail = true;
    if (fail) break;
    const int nchunk = bcast[2];

    DTW_TICK(3);

    // Back-trace segments (section 3): the shortest of 4 / 8 / 16 rows that keeps the candidates (about cells /
    // rows-per-segment) within one round of the workgroup's threads -- the walks are the dependent part, shorter is faster
    const int kSeg = ncell_lvl <= kSegMin * kThreads ? kSegMin : (ncell_lvl <= 2 * kSegMin * kThreads ? 2 * kSegMin : kSegMax);
    const int G = (ltx + kSeg - 1) / kSeg;  // segment g = rows [g*kSeg, min((g+1)*kSeg, ltx))
    // candidate-table offsets: one entry per cell of every segment's bottom row.  They depend on the windows only:
    // one of the cost wavefronts computes them while the last chunk is swept (it has no further chunk to prepare).
    auto candidate_offsets = [&]() {
      const int gpl = (G + 63) / 64;
      const int g0 = lane * gpl < G ? lane * gpl : G;
      const int g1 = g0 + gpl < G ? g0 + gpl : G;
      int sum = 0;
      for (int g = g0; g < g1; ++g) {
        const int bot = ((g + 1) * kSeg < ltx ? (g + 1) * kSeg : ltx) - 1;
        sum += HI(bot) - LO(bot) + 1;
      }
      int total;
      int run = wave_excl_scan(sum, lane, &total);
      for (int g = g0; g < g1; ++g) {
        const int bot = ((g + 1) * kSeg < ltx ? (g + 1) * kSeg : ltx) - 1;
        segoff[g] = run;
        run += HI(bot) - LO(bot) + 1;
      }
      if (lane == 0) segoff[G] = total;
    };

    // ---- 2d. DP: wavefront 0 sweeps chunk c while the other wavefronts prepare the costs of chunk c+1 ----
    int prevlo = -1, prevhi = -1;  // the virtual row -1 has the single cell (-1, -1)
    int hw_next = 0;               // next free back-pointer halfword
    double last_val = INFINITY;
    for (int c = 0; c < nchunk; ++c) {
      const int i0 = (int)cstart[c], R = (int)cstart[c + 1] - i0, base = OFF(i0);
      double *dcur = dchunk + (c & 1) * dstride;
      double *dnxt = dchunk + ((c + 1) & 1) * dstride;  // its head receives the row handed to chunk c+1
      const int last_lo = LO(i0 + R - 1), last_hi = HI(i0 + R - 1);
      if (!w0 && c + 1 < nchunk) chunk_costs(c + 1, dnxt, 64, kThreads - 64);
      if (c + 1 == nchunk && (tid >> 6) == 1) candidate_offsets();
      // anti-diagonal sweep (wavefront 0): lane r >= 1 owns row i0 + r - 1, lane 0 feeds the row above;
      // at step s every lane handles column s - lane of its row
      if (w0) {
      const bool feeder = lane == 0;
      const bool real = lane >= 1 && lane <= R;
      const bool is_last = lane == R;  // hands its row to the next chunk
      const int i = i0 + lane - 1;
      // Everything the sweep touches in LDS is addressed by absolute LDS byte addresses (address_space(3) pointers:
      // no base add per access) that advance by running offsets.
      typedef __attribute__((address_space(3))) unsigned char lds_u8;
      typedef __attribute__((address_space(3))) unsigned short lds_u16;
      typedef __attribute__((address_space(3))) double lds_f64;
      const int lbase = (int)(unsigned)(uintptr_t)(lds_u8 *)smem;
      auto lds_addr = [&](const void *q) { return lbase + (int)((const unsigned char *)q - smem); };
      // steps s = s0 .. s1 in blocks of 8, padded at the FRONT (all-+INF lead-in): the sweep ends exactly on the last
      // row's last column.  Two steps of lead-in at least: the feeder emits columns lo-1 and lo first.
      const int s1 = last_hi + R;
      const int nblocks = (s1 - __builtin_amdgcn_readfirstlane(LO(i0)) + 9) >> 3;
      const int s0 = s1 + 1 - 8 * nblocks;
      int mylo = 0, width = 0;
      int src = lds_addr(dcur + kSlack);  // byte address of the local cost of the row's column lo
      if (feeder) {
        mylo = prevlo;
        width = prevhi - prevlo + 1;
      } else if (real) {
        const unsigned lh = rw[2 * i];
        mylo = (int)(lh & 0xffffu);
        width = (int)(lh >> 16) - mylo + 1;
        src = lds_addr(dcur + feed_cells(i0) + (OFF(i) - base) + 4 * (lane - 1) + 2);
      }
      const int c0 = s0 - lane - mylo;  // column of step 0 relative to the row's window; <= -2 for the rows of the chunk
      // back-pointer codes, one halfword per block: the blocks that touch the row's window (steps k_in .. k_in +
      // width - 1) get consecutive halfwords from H on, every other block -- and all blocks of the feeder and the idle
      // lanes -- goes to the lane's throw-away halfword
      const int k_in = -c0;
      const int nhw = real ? ((k_in + width - 1) >> 3) - (k_in >> 3) + 1 : 0;
      int hw_total;
      const int H = hw_next + wave_excl_scan(nhw, lane, &hw_total);
      hw_next += hw_total;
      if (hw_next > p.hwcap) {  // uniform: the level does not fit (optimistic capacities); nothing is written
        if (lane == 0) bcast[7] = 1;
      } else {
      const int hw_dummy = lds_addr(bp16 + bp_dummy + lane);
      int hw_run = lds_addr(bp16) + 2 * H;  // halfword of block q_rel = 0
      int q_rel = real ? -(k_in >> 3) : 0;  // block index relative to the first block that touches the window
      if (real) rw[2 * i + 1] = (unsigned)(8 * H + (k_in & 7));  // from here on: the row's cell-index base
      // local costs, read in pairs (steps 2m, 2m + 1) one block ahead; the pair address is clamped to the row's
      // +INF frame: [f_lo, f_hi] = the pairs (lo-2, lo-1) .. (hi+1, hi+2).  Pair q of a block:
      // med3(f_run + 16 q, f_lo, f_hi) = med3(f_run, f_lo - 16 q, f_hi - 16 q) + 16 q, and the + 16 q rides in the
      // instruction's offset field.
      const int f_lo = src - 16, f_hi = (feeder || real) ? src + 8 * width : src - 16;
      int f_run = src + 8 * c0;
      auto load_block = [&](double (&d)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = med3_i32(f_run, f_lo - 16 * q, f_hi - 16 * q);
          const lds_f64 *pp = (const lds_f64 *)(uintptr_t)(unsigned)a;
          d[2 * q] = pp[2 * q];
          d[2 * q + 1] = pp[2 * q + 1];
        }
        f_run += 64;
      };
      // the row handed to the next chunk (written by the last lane only): a block whose first column lies in
      // [-kSlack, width] is written as it is (what overhangs the window is +INF by itself and lands in the slack),
      // any other block -- and every block of the other lanes -- goes to the lane's throw-away slots
      const int d_m8 = lds_addr(dnxt);  // cell -kSlack
      const int d_dummy = lds_addr(ddummy + lane);
      const unsigned wlim = is_last ? (unsigned)(width + kSlack + 1) : 0u;
      int x_run = c0 + kSlack;
      double pub = INFINITY;  // this lane's D at the column of the previous step (+INF outside the window)
      double upp = INFINITY;  // row above at the previous column (the diagonal)
      unsigned acc = 0u;
      // One block = 8 steps, 13 vector instructions each: the DPP hand-down (2), three adds, two min, two
      // compare + shift-in, one hand-over store, and half a cost-pair load.  The loop-carried chain is
      // DPP -> add -> min (the left / diagonal minimum is formed beside it).  The feeder receives +0.0 from the DPP
      // (bound_ctrl), which makes it replay its stored row: up + dt = diag + dt = dt exactly and left + dt >= dt (D >= 0).
      // Codes (TIE_FIRST_MIN): bit a = "diagonal < left", then bit b = "min(left, diagonal) < up": up unless b, else
      // diagonal if a, else left -- the first minimum in the order up, left, diagonal, as the oracle's three compares.
      // (TIE_DIAG_LAST: the same two compares with their operands exchanged, see bp_code.)
      auto block = [&](const double (&dt)[8], auto store_tag) {
        constexpr bool STORE = decltype(store_tag)::value;
        const int dbase = ((unsigned)x_run < wlim) ? d_m8 + 8 * x_run : d_dummy;
        lds_f64 *dw = (lds_f64 *)(uintptr_t)(unsigned)dbase;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const double up = wave_shr1z(pub);
          const double cl = __dadd_rn(pub, dt[k8]), cd = __dadd_rn(upp, dt[k8]);
          const double x = vmin_f64(cl, cd);  // off the cross-lane chain: both operands are a step old
          acc = TIE == MLPG_HIP_TIE_FIRST_MIN ? shift_in_lt(acc, cd, cl) : shift_in_lt(acc, cl, cd);
          const double cu = __dadd_rn(up, dt[k8]);
          const double best = vmin_f64(cu, x);
          acc = TIE == MLPG_HIP_TIE_FIRST_MIN ? shift_in_lt(acc, x, cu) : shift_in_lt(acc, cu, x);
          if (STORE) dw[k8] = best;
          upp = up;
          pub = best;
        }
        *(lds_u16 *)(uintptr_t)(unsigned)((unsigned)q_rel < (unsigned)nhw ? hw_run + 2 * q_rel : hw_dummy) = (unsigned short)acc;
        q_rel += 1;
        x_run += 8;
      };
      // The last lane reaches its window only in the final blocks of the chunk: the blocks before (an even number of
      // them) run without the hand-over stores -- a sixth of the sweep's time when every block carried them.
      const int c0_last = s0 - R - last_lo;  // the last lane's c0 (uniform)
      int b_st = (-c0_last - kSlack) >> 3;   // its first block whose columns reach -kSlack: floor((-kSlack - c0) / 8)
      b_st = b_st < 0 ? 0 : (b_st > nblocks ? nblocks : b_st) & ~1;
      double dA[8], dB[8];
      load_block(dA);
      for (int b = 0; b < b_st; b += 2) {
        load_block(dB);
        block(dA, std::false_type());
        load_block(dA);
        block(dB, std::false_type());
      }
      for (int b = b_st; b < nblocks; b += 2) {
        load_block(dB);
        block(dA, std::true_type());
        if (b + 1 >= nblocks) break;
        load_block(dA);
        block(dB, std::true_type());
      }
      if (is_last) {  // +INF frame of the row handed to the next chunk
        dnxt[kSlack - 2] = INFINITY;
        dnxt[kSlack - 1] = INFINITY;
        dnxt[kSlack + width] = INFINITY;
        dnxt[kSlack + width + 1] = INFINITY;
      }
      // value of the chunk's last cell (bottom-right corner of the level if this is the last chunk)
      last_val = __shfl(pub, R);
      if (lane == 0) *(double *)(bcast + 12) = last_val;  // for every wavefront: the back-trace's segments are dealt to all threads
      }  // fits
      }  // w0
      prevlo = last_lo;
      prevhi = last_hi;
      DTW_TICK(4);
      __syncthreads();
      DTW_TICK(2);  // the sweeper waiting for the next chunk's costs
      if (bcast[7]) break;
    }
    if (bcast[7]) fail = true;
    if (fail) break;
    last_val = *(const double *)(bcast + 12);  // (written before the last chunk's barrier)
    level_cost = (prevhi == lty - 1) ? last_val : INFINITY;

    // ---- 3. back-trace ----
    // level 0 writes the path straight to the output arrays (from position 0), the other levels into LDS (right-aligned)
    const int wcap = k == 0 ? pcap : pcapL;
    DTW_TICK(7);
    const int ntask = segoff[G];
    // candidate tables in the (now idle) cost buffers: cnts, then `nlev` hop tables of ntask entries
    // each; hop table k maps a candidate to the candidate reached 2^k segments further up
    int nlev = 1;
    while ((1 << nlev) < G) ++nlev;
    unsigned short *cnts = (unsigned short *)dchunk;
    unsigned short *hop = cnts + ntask;
    constexpr unsigned kInvalid = 0xffffu, kTerminal = 0xfffeu;
    if ((size_t)ntask * (size_t)(nlev + 1) * sizeof(unsigned short) <= sizeof(double) * 2 * (size_t)dstride &&
        ntask < 0xfff0 && G < segcap) {
      // pass 1: every (segment, entry column) candidate walks to the top of its segment
      for (int task = tid; task < ntask; task += kThreads) {
        int a = 0, b = G;
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (segoff[mid] <= task) a = mid; else b = mid;
        }
        const int bot = ((a + 1) * kSeg < ltx ? (a + 1) * kSeg : ltx) - 1;
        int nc;
        const int ex = dtw_walk<false, TIE>(rinfo, bp16, pth_i, pth_j, bot, LO(bot) + (task - segoff[a]), a * kSeg, 0, &nc);
        cnts[task] = (unsigned short)nc;
        unsigned nx = kInvalid;
        if (a == 0) {
          if (ex == -1) nx = kTerminal;
        } else if (ex >= 0) {
          const int up = a * kSeg - 1;  // bottom row of the segment above
          if (ex >= LO(up) && ex <= HI(up)) nx = (unsigned)(segoff[a - 1] + ex - LO(up));
        }
        hop[task] = (unsigned short)nx;
      }
      if (tid == 0) bcast[1] = 1;
      __syncthreads();
      DTW_TICK(8);
      // hop tables by doubling
      for (int q = 1; q < nlev; ++q) {
        const unsigned short *hp = hop + (size_t)(q - 1) * ntask;
        unsigned short *hn = hop + (size_t)q * ntask;
        for (int task = tid; task < ntask; task += kThreads) {
          const unsigned j = hp[task];
          hn[task] = j >= kTerminal ? (unsigned short)j : hp[j];
        }
        __syncthreads();
      }
      // the candidate of every segment on the path from the bottom-right corner: segment g is
      // G-1-g hops above the start
      {
        const int last = ltx - 1;
        const bool start_ok = (level_cost < INFINITY) && lty - 1 >= LO(last) && lty - 1 <= HI(last);
        for (int g = tid; g < G; g += kThreads) {
          unsigned t = start_ok ? (unsigned)(segoff[G - 1] + lty - 1 - LO(last)) : kInvalid;
          const int h = G - 1 - g;
          for (int q = 0; q < nlev && t < kTerminal; ++q)
            if ((h >> q) & 1) t = hop[(size_t)q * ntask + t];
          bool good = t < kTerminal;
          if (good && g == 0) good = hop[t] == kTerminal;  // the path must end in (-1, -1)
          if (good) {
            segent[g] = (int)t - segoff[g];
            segend[g] = (int)cnts[t];
          } else {
            segend[g] = 0;
            bcast[1] = 0;
          }
        }
      }
      __syncthreads();
      // positions: inclusive prefix sum of the per-segment cell counts (wavefront 0)
      if (w0) {
        const int gpl = (G + 63) / 64;
        const int g0 = lane * gpl < G ? lane * gpl : G;
        const int g1 = g0 + gpl < G ? g0 + gpl : G;
        int sum = 0;
        for (int g = g0; g < g1; ++g) sum += segend[g];
        int total;
        int run = wave_excl_scan(sum, lane, &total);
        const int first = k == 0 ? 0 : wcap - total;
        for (int g = g0; g < g1; ++g) {
          run += segend[g];
          segend[g] = first + run;
        }
        if (lane == 0) {
          bcast[0] = first;
          bcast[5] = total;
          if (total > wcap) bcast[1] = 0;
        }
      }
      __syncthreads();
      DTW_TICK(9);
      // pass 2: every segment writes its piece of the path
      if (bcast[1]) {
        for (int g = tid; g < G; g += kThreads) {
          const int bot = ((g + 1) * kSeg < ltx ? (g + 1) * kSeg : ltx) - 1;
          int nc;
          if (k == 0) (void)dtw_walk<true, TIE>(rinfo, bp16, out_i, out_j, bot, LO(bot) + segent[g], g * kSeg, segend[g], &nc);
          else (void)dtw_walk<true, TIE>(rinfo, bp16, pth_i, pth_j, bot, LO(bot) + segent[g], g * kSeg, segend[g], &nc);
        }
      }
    } else if (w0) {
      // sequential fallback (very wide windows): the walk is wave-uniform, so the loaded values are
      // made scalar (readfirstlane) and the control flow runs on the scalar unit.  First pass counts, second writes.
      auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
      int ok = uni((level_cost < INFINITY) ? 1 : 0), total = 0, first = 0;
      for (int pass = 0; pass < 2 && ok; ++pass) {
        int bi = ltx - 1, bj = lty - 1, pos = first + total;
        int cnt = 0;
        while (ok) {
          const int rl = uni(LO(bi)), rh = uni(HI(bi)), cb = uni(OFF(bi));
          if (bj < rl || bj > rh) { ok = 0; break; }
          bool up = false;
          while (!up) {  // cells of this row on the path
            if (cnt >= wcap || bj < rl) { ok = 0; break; }
            const int code = uni((int)bp_code<TIE>(bp16, cb + bj - rl));
            ++cnt;
            if (pass == 1) {
              --pos;
              if (lane == 0) {
                if (k == 0) {
                  out_i[pos] = bi;
                  out_j[pos] = bj;
                } else {
                  pth_i[pos] = (unsigned short)bi;
                  pth_j[pos] = (unsigned short)bj;
                }
              }
            }
            if (code != 0) bj -= 1;
            up = code != 1;
          }
          if (!ok) break;
          bi -= 1;
          if (bi < 0) {
            if (bj != -1) ok = 0;
            break;
          }
        }
        if (pass == 0) {
          total = cnt;
          first = k == 0 ? 0 : wcap - total;
        }
      }
      if (lane == 0) {
        bcast[0] = first;
        bcast[5] = total;
        bcast[1] = ok;
      }
    }
    __syncthreads();
    pstart = bcast[0];
    pn = bcast[5];
    if (!bcast[1]) fail = true;
    __syncthreads();
    DTW_TICK(5);
  }

  if (fail) {
    if (tid == 0) {
      p.path_len[n] = p.tier == 1 ? -1 : 0;
      p.cost[n] = NAN;
    }
    return;
  }
  if (tid == 0) {
    p.path_len[n] = pn;
    p.cost[n] = level_cost;
  }
#ifdef MLPG_DTW_TIMING
  DTW_TICK(6);
  __syncthreads();
  if (tid == 0)
    for (int q = 0; q < 11; ++q) out_i[pcap - 12 + q] = (int)(tq[q] >> 4);   // profiling build: cycles / 16 in the tail of path_i
#endif
}

size_t lds_bytes(int Tx, int Ty, const DtwParams &p) {
  size_t b = 0;
  b += sizeof(double) * (2 * ((size_t)p.chunkcap + 4 * kRows + 4) + kHandDummy);
  b += sizeof(unsigned long long) * (size_t)(Tx + 1);
  b += sizeof(int) * (2 * kMaxLevels + 16 + 3 * (size_t)(Tx / kSegMin + 3));
  b += sizeof(unsigned short) * (2 * (size_t)(Tx / 2 + 2) + 2 * (size_t)p.pcap_lds);
  b += sizeof(unsigned short) * ((size_t)p.hwcap + 64) + 16;  // back-pointer halfwords + one throw-away halfword per lane
  return (b + 15) & ~(size_t)15;
}

template <int kThreads, int TIE>
int launch_one(hipStream_t s, const DtwParams &p, int N, size_t lds) {
  constexpr int kMaxDevices = 16;
  static size_t attr_set[kMaxDevices] = {};  // largest dynamic-LDS size the attribute was set to, per device
  static std::mutex attr_mu;
  std::lock_guard<std::mutex> attr_lk(attr_mu);
  int device = 0;
  MLPG_HIP_CHECK(hipGetDevice(&device));
  if (device < 0 || device >= kMaxDevices || attr_set[device] < lds) {
    MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)fastdtw_kernel<kThreads, TIE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    if (device >= 0 && device < kMaxDevices) attr_set[device] = lds;
  }
  hipLaunchKernelGGL((fastdtw_kernel<kThreads, TIE>), dim3(N), dim3(kThreads), lds, s, p);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace

int launch_fastdtw(hipStream_t s, int device, const double *X, const double *Y, const int32_t *lenx,
                   const int32_t *leny, int N, int Tx, int Ty, int D, int radius, int dist_kind, double dist_scale,
                   int tie_rule, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *cost) {
  if (tie_rule != MLPG_HIP_TIE_FIRST_MIN && tie_rule != MLPG_HIP_TIE_DIAG_LAST) {
    set_error("fastdtw: unknown tie rule %d", tie_rule);
    return MLPG_HIP_EINVAL;
  }
  if (dist_kind < MLPG_HIP_DIST_L2 || dist_kind > MLPG_HIP_DIST_SCALED_SQL2_NP) {
    set_error("fastdtw: unknown local distance %d", dist_kind);
    return MLPG_HIP_EINVAL;
  }
  if (dist_kind != MLPG_HIP_DIST_L2 && D > 128) {
    set_error("fastdtw: the MLPG_HIP_DIST_*_NP distances reproduce numpy's summation order up to 128 feature dims (got %d)", D);
    return MLPG_HIP_EINVAL;
  }
  if (Tx > 65000 || Ty > 65000) {
    set_error("fastdtw: sequences longer than 65000 frames are not supported");
    return MLPG_HIP_EINVAL;
  }
  DtwParams p;
  p.X = X; p.Y = Y; p.lenx = lenx; p.leny = leny;
  p.N = N; p.Tx = Tx; p.Ty = Ty; p.D = D; p.radius = radius;
  p.path_i = path_i; p.path_j = path_j; p.path_len = path_len; p.cost = cost;
  p.dist_kind = dist_kind;
  p.dist_scale = dist_scale;
  p.pyr_stride = (size_t)(Tx + Ty) * D;
  p.pcap_lds = (Tx + Ty) / 2 + 2;
  p.tier = 0;
  // BOUNDS: window cells per level <= (4r+2)(tx+ty); the coarsest level runs a full DTW with one side
  // <= r+1.  Back-pointer halfwords: per row <= ((width + 6) >> 3) + 1 <= width / 8 + 1.75.  A cost buffer must take
  // the widest handed-over row and the widest row (both <= Ty) with the slack.
  long cc = (long)(4 * radius + 2) * (Tx + Ty) + 64;
  const long full = (long)Tx * Ty;
  if (full < cc) cc = full + 64;
  p.cellcap = (int)cc;
  p.hwcap = (int)(cc / 8 + 2 * (long)Tx + 2);
  p.chunkcap = 2 * Ty + 2 * kSlack > 1024 ? 2 * Ty + 2 * kSlack : 1024;
  const size_t lds = lds_bytes(Tx, Ty, p);
  if (lds > 160 * 1024) {
    set_error("fastdtw: Tx=%d, Ty=%d, D=%d, radius=%d needs %zu bytes of LDS (> 160 KiB)", Tx, Ty, D, radius, lds);
    return MLPG_HIP_EINVAL;
  }
  constexpr size_t kTicketBytes = 2048 * sizeof(int);
  p.cu_tickets = (int *)scratch(device, s, 1, kTicketBytes + sizeof(double) * p.pyr_stride * (size_t)N);
  if (!p.cu_tickets) return MLPG_HIP_ENOMEM;
  p.pyr = (double *)((char *)p.cu_tickets + kTicketBytes);
  // Many pairs: an optimistic first launch with capacities for the usual window shapes (rows up to ~20 cells) and
  // 256 threads per pair -- four pairs resident per CU instead of two -- and a second launch with the bounds for
  // the pairs that did not fit (they are marked path_len = -1; normally none, the launch is a few microseconds).
  const int cus = 256;
  DtwParams q = p;
  q.tier = 1;
  q.chunkcap = 704;
  q.hwcap = (int)(2.5 * Tx) + 64;
  if (q.hwcap > p.hwcap) q.hwcap = p.hwcap;
  const size_t lds_q = lds_bytes(Tx, Ty, q);
#ifdef MLPG_DTW_MEASURE  // measurement builds only (tools/dbg/dtw_scaling.py, dtw_tiers.py): 1 two launches, 2 one
  static const int force = [] { const char *e = getenv("MLPG_HIP_DTW_FORCE"); return e ? atoi(e) : 0; }();
#else
  constexpr int force = 0;
#endif
  if (force != 2 && (N > 2 * cus || force == 1) && q.chunkcap < p.chunkcap && lds_q <= 40 * 1024) {
    if (int rc = tie_rule == MLPG_HIP_TIE_FIRST_MIN ? launch_one<256, MLPG_HIP_TIE_FIRST_MIN>(s, q, N, lds_q)
                                                     : launch_one<256, MLPG_HIP_TIE_DIAG_LAST>(s, q, N, lds_q))
      return rc;
#ifdef MLPG_DTW_MEASURE  // leaves the pairs that overflowed the optimistic capacities at path_len = -1
    static const bool first_only = getenv("MLPG_HIP_DTW_FIRST_LAUNCH_ONLY") != nullptr;
    if (first_only) return 0;
#endif
    p.tier = 2;
  }
  return tie_rule == MLPG_HIP_TIE_FIRST_MIN ? launch_one<512, MLPG_HIP_TIE_FIRST_MIN>(s, p, N, lds)
                                            : launch_one<512, MLPG_HIP_TIE_DIAG_LAST>(s, p, N, lds);
}

}  // namespace mlpg
