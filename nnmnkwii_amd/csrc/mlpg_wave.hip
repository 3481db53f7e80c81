// Wave-per-system MLPG kernels (the fast path; algo = MLPG_HIP_ALGO_WAVE / AUTO).
//
// One 64-lane wavefront solves one (utterance, static dim) system; a workgroup
// of G wavefronts takes G consecutive static dims of one utterance so that the
// (T, D) row-major inputs can be staged through LDS in runs of G elements per
// frame.  Windows must have extents l, u <= 1 (half-bandwidth <= 2): that is
// the static/delta/delta-delta family; anything else goes to the generic kernel.
//
// Per system (T frames, lane p owns the chunk of M consecutive frames
// [pM, pM+M), 64*M >= T, frames >= T are identity rows):
//   1. assemble the pentadiagonal P (diagonal + 2 sub-diagonals) and b for the
//      chunk in REGISTERS, one window at a time from two LDS tiles (variance,
//      mean) that all G wavefronts load cooperatively and coalesced;
//   2. substructuring: each lane eliminates its M-2 interior frames (LDL^T,
//      sequential, M-2 steps) carrying the two "left spike" columns that couple
//      it to the previous lane's last 2 frames; the elimination runs on into the
//      lane's own last 2 frames (its separator), which yields the Schur
//      complement: a block-tridiagonal SPD system with 2x2 blocks over the 64
//      separators;
//   3. that reduced system is solved across the lanes by parallel cyclic
//      reduction (6 steps, data exchanged with cross-lane shuffles);
//   4. each lane back-substitutes its interior (two short sweeps);
//   5. the trajectory goes back through LDS and is stored coalesced.
// The factor never leaves the register file; HBM traffic is the algorithmic
// minimum (read means + variances once, write the trajectory once).
//
// Reference semantics reproduced (paramgen/_mlpg.py:92-199): tau = 1/var in the
// input dtype, dynamic-window precisions zeroed on the first/last mw frames,
// float64 arithmetic, output cast to the input dtype, status = index of the
// first non-positive pivot of the NATURAL-order Cholesky (linalg.pyx:79-82;
// found by a sequential re-scan on the rare failing system).
#include "assemble.h"

namespace mlpg {
namespace {

constexpr int kSkew = 1;  // one padding slot per chunk in the LDS tiles (bank-conflict-free)

template <int M>
struct Geo {
  static constexpr int G = (M <= 16) ? 8 : 4;        // systems (wavefronts) per workgroup
  static constexpr int NT = G * 64;                  // threads per workgroup
  static constexpr int TPAD = 64 * (M + kSkew) + 2;  // tile pitch per system; % 16 == 2
  static constexpr int LOG2M = (M == 4) ? 2 : (M == 8) ? 3 : (M == 16) ? 4 : 5;
};

template <int M>
__device__ __forceinline__ int tile_idx(int t) {
  return t + (t >> Geo<M>::LOG2M) * kSkew;
}

// Cooperative, coalesced load of one feature column group into an LDS tile:
//   tile[g][t] = src[t * row_stride + g],  g < gvalid, t < T
template <int M, typename TIN>
__device__ __forceinline__ void load_tile(TIN *__restrict__ tile, const TIN *__restrict__ src, int row_stride, int T,
                                          int gvalid, int tid) {
  constexpr int G = Geo<M>::G, NT = Geo<M>::NT, TP = Geo<M>::TPAD;
  const int g = tid % G;
  const int total = T * G;
  constexpr int U = 8;
  for (int e0 = tid; e0 < total; e0 += NT * U) {
    TIN v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = e0 + k * NT;
      const int t = e / G;
      v[k] = (e < total && g < gvalid) ? src[(size_t)t * row_stride + g] : (TIN)0;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int e = e0 + k * NT;
      const int t = e / G;
      if (e < total) tile[g * TP + tile_idx<M>(t)] = v[k];
    }
  }
}

// Cooperative, coalesced store of an LDS tile: dst[t * row_stride + g] = tile[g][t] for
// t < T, zero for T <= t < Tmax.
template <int M, typename TOUT>
__device__ __forceinline__ void store_tile(const TOUT *__restrict__ tile, TOUT *__restrict__ dst, int row_stride, int T,
                                           int Tmax, int gvalid, int tid) {
  constexpr int G = Geo<M>::G, NT = Geo<M>::NT, TP = Geo<M>::TPAD;
  const int g = tid % G;
  if (g >= gvalid) return;
  const int total = Tmax * G;
  for (int e = tid; e < total; e += NT) {
    const int t = e / G;
    dst[(size_t)t * row_stride + g] = t < T ? tile[g * TP + tile_idx<M>(t)] : (TOUT)0;
  }
}

__device__ __forceinline__ double shfl_up_d(double v, int delta) { return __shfl_up(v, delta); }
__device__ __forceinline__ double shfl_dn_d(double v, int delta) { return __shfl_down(v, delta); }

template <int M, typename TIN, typename TOUT, bool BWD>
__global__ __launch_bounds__(Geo<M>::NT) void wave_kernel(Problem p, WinSet ws, int ngrp, int nslots) {
  constexpr int G = Geo<M>::G, TP = Geo<M>::TPAD;
  constexpr int n = M - 2;  // interior frames per lane; frames n, n+1 form the lane's separator
  extern __shared__ __align__(16) unsigned char smem[];
  // two tiles sized for 8-byte elements; reused as variance/mean input tiles and as the output tile
  double *tileA_raw = (double *)smem;
  double *tileB_raw = tileA_raw + G * TP;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // XCD-aware decode: workgroup w runs on XCD w % 8; consecutive slots of one XCD are the
  // static-dim groups of one utterance, so sibling groups (which share 128-byte lines of the
  // row-major input) hit the same L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= nslots) return;
  const int b = (slot / ngrp) * 8 + xcd, dgrp = slot % ngrp;
  if (b >= p.B) return;
  const int sd = p.sd, D = p.D, Tmax = p.Tmax;
  const int d0 = dgrp * G, d = d0 + wv;
  const int gvalid = sd - d0 < G ? sd - d0 : G;
  const bool sys_valid = d < sd;
  int T = p.lengths ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  const int mw = ws.mw, nw = ws.nw, var_mode = p.var_mode;

  const TIN *mean_b = BWD ? nullptr : (const TIN *)p.mean + (size_t)b * Tmax * D;
  const TIN *var_b = (const TIN *)p.var;
  if (var_mode == MLPG_HIP_VAR_FRAME) var_b += (size_t)b * Tmax * D;
  const TIN *gout_b = BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * sd : nullptr;

  const int f0 = lane * M;  // first frame of this lane's chunk

  // ---- 1. assembly: Pd[i] = P[f,f], P1[i] = P[f+1,f], P2[i] = P[f+2,f], rhs[i], f = f0 + i ----
  double Pd[M], P1[M], P2[M], rhs[M];
#pragma unroll
  for (int i = 0; i < M; ++i) Pd[i] = P1[i] = P2[i] = rhs[i] = 0.0;

  TIN *tileV = (TIN *)tileA_raw, *tileM = (TIN *)tileB_raw;

  if (BWD) {
    load_tile<M, TIN>(tileM, gout_b + d0, sd, T, gvalid, tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int t = f0 + i;
      rhs[i] = (t < T) ? (double)tileM[wv * TP + tile_idx<M>(t)] : 0.0;
    }
    __syncthreads();
  }

  for (int w = 0; w < nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *cw = ws.c + ws.off[w];
    const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;  // W[t,t-1], W[t,t], W[t,t+1]
    if (var_mode == MLPG_HIP_VAR_FRAME) load_tile<M, TIN>(tileV, var_b + w * sd + d0, D, T, gvalid, tid);
    if (!BWD) load_tile<M, TIN>(tileM, mean_b + w * sd + d0, D, T, gvalid, tid);
    __syncthreads();
    double tau_glob = 1.0;
    if (var_mode == MLPG_HIP_VAR_GLOBAL && sys_valid) tau_glob = recip_in_dtype<TIN>(var_b[w * sd + d]);
    const double c00 = c0 * c0, cpp = cp * cp, cmm = cm * cm, cp0 = cp * c0, c0m = c0 * cm, cpm = cp * cm;
    // one pass over the chunk plus a halo frame on each side: frame t feeds rows t-1, t, t+1
#pragma unroll
    for (int i = -1; i <= M; ++i) {
      const int t = f0 + i;
      double tau = 0.0, tm = 0.0;
      const bool live = t >= 0 && t < T && !(w != 0 && (mw == 0 || t < mw || t >= T - mw));
      if (live) {
        tau = (var_mode == MLPG_HIP_VAR_FRAME) ? recip_in_dtype<TIN>(tileV[wv * TP + tile_idx<M>(t)]) : tau_glob;
        if (!BWD) tm = tau * (double)tileM[wv * TP + tile_idx<M>(t)];
      }
      if (i >= 0 && i < M) {  // row f = t:  t == f
        Pd[i] += c00 * tau;
        P1[i] += cp0 * tau;   // P[f+1,f] gets W[f,f+1] W[f,f] tau[f]
        if (!BWD) rhs[i] += c0 * tm;
      }
      if (i + 1 >= 0 && i + 1 < M) {  // row f = t+1:  t == f-1
        Pd[i + 1] += cpp * tau;
        if (!BWD) rhs[i + 1] += cp * tm;
      }
      if (i - 1 >= 0 && i - 1 < M) {  // row f = t-1:  t == f+1
        Pd[i - 1] += cmm * tau;
        P1[i - 1] += c0m * tau;  // W[f+1,f+1] W[f+1,f] tau[f+1]
        P2[i - 1] += cpm * tau;  // W[f+1,f+2] W[f+1,f] tau[f+1]
        if (!BWD) rhs[i - 1] += cm * tm;
      }
    }
    __syncthreads();
  }
  // matrix edges: rows >= T are identity rows, entries that would leave the T x T matrix vanish
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int f = f0 + i;
    if (f >= T) {
      Pd[i] = 1.0;
      P1[i] = P2[i] = rhs[i] = 0.0;
    } else {
      if (f + 1 >= T) P1[i] = 0.0;
      if (f + 2 >= T) P2[i] = 0.0;
    }
  }

  // ---- 2. interior elimination with left spikes ----
  // coupling of this chunk's first two frames to the previous lane's separator (frames f0-2, f0-1)
  double ca = shfl_up_d(P2[M - 2], 1);  // P[f0,   f0-2]
  double cb = shfl_up_d(P1[M - 1], 1);  // P[f0,   f0-1]
  double cc = shfl_up_d(P2[M - 1], 1);  // P[f0+1, f0-1]
  if (lane == 0) ca = cb = cc = 0.0;

  bool bad = false;
  double t00 = 0.0, t01 = 0.0, t11 = 0.0, h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    const double dd = Pd[i];
    bad |= (dd <= 0.0);
    const double dinv = 1.0 / dd;
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
  }
  // run the elimination on into the separator rows (frames n, n+1 of the chunk)
  rhs[n] -= l1p * g1 + l2pp * g2;
  rhs[n + 1] -= l2p * g1;
  // coupling block of this separator (rows) to the previous one (columns a, b)
  double L11 = -(l1p * va1 + l2pp * va2), L12 = -(l1p * vb1 + l2pp * vb2);
  double L21 = -(l2p * va1), L22 = -(l2p * vb1);
  // Schur contributions of the NEXT lane's left spikes land on this lane's separator block
  double D11 = Pd[n], D12 = P1[n], D22 = Pd[n + 1];
  double F1 = rhs[n], F2 = rhs[n + 1];
  {
    const double n00 = shfl_dn_d(t00, 1), n01 = shfl_dn_d(t01, 1), n11 = shfl_dn_d(t11, 1);
    const double nh0 = shfl_dn_d(h0, 1), nh1 = shfl_dn_d(h1, 1);
    if (lane < 63) {
      D11 -= n00; D12 -= n01; D22 -= n11;
      F1 -= nh0; F2 -= nh1;
    }
  }
  // coupling to the next separator = transpose of the next lane's block
  double U11, U12, U21, U22;
  {
    const double a11 = shfl_dn_d(L11, 1), a12 = shfl_dn_d(L12, 1), a21 = shfl_dn_d(L21, 1), a22 = shfl_dn_d(L22, 1);
    U11 = a11; U12 = a21; U21 = a12; U22 = a22;
    if (lane == 63) U11 = U12 = U21 = U22 = 0.0;
  }
  if (lane == 0) L11 = L12 = L21 = L22 = 0.0;

  // ---- 3. block-tridiagonal reduced system over the 64 separators: parallel cyclic reduction ----
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const double det = D11 * D22 - D12 * D12;
    bad |= (D11 <= 0.0) | (det <= 0.0);
    const double idet = 1.0 / det;
    const double I11 = D22 * idet, I12 = -D12 * idet, I22 = D11 * idet;
    const bool hasm = lane - s >= 0, hasp = lane + s < 64;
    // rows lane-s and lane+s
    double mI11 = __shfl_up(I11, s), mI12 = __shfl_up(I12, s), mI22 = __shfl_up(I22, s);
    double mU11 = __shfl_up(U11, s), mU12 = __shfl_up(U12, s), mU21 = __shfl_up(U21, s), mU22 = __shfl_up(U22, s);
    double mL11 = __shfl_up(L11, s), mL12 = __shfl_up(L12, s), mL21 = __shfl_up(L21, s), mL22 = __shfl_up(L22, s);
    double mF1 = __shfl_up(F1, s), mF2 = __shfl_up(F2, s);
    double pI11 = __shfl_down(I11, s), pI12 = __shfl_down(I12, s), pI22 = __shfl_down(I22, s);
    double pU11 = __shfl_down(U11, s), pU12 = __shfl_down(U12, s), pU21 = __shfl_down(U21, s), pU22 = __shfl_down(U22, s);
    double pL11 = __shfl_down(L11, s), pL12 = __shfl_down(L12, s), pL21 = __shfl_down(L21, s), pL22 = __shfl_down(L22, s);
    double pF1 = __shfl_down(F1, s), pF2 = __shfl_down(F2, s);
    if (!hasm) { mI11 = mI12 = mI22 = 0.0; mU11 = mU12 = mU21 = mU22 = 0.0; mL11 = mL12 = mL21 = mL22 = 0.0; mF1 = mF2 = 0.0; }
    if (!hasp) { pI11 = pI12 = pI22 = 0.0; pU11 = pU12 = pU21 = pU22 = 0.0; pL11 = pL12 = pL21 = pL22 = 0.0; pF1 = pF2 = 0.0; }
    // alpha = -L * inv(D[lane-s]),  beta = -U * inv(D[lane+s])
    const double A11 = -(L11 * mI11 + L12 * mI12), A12 = -(L11 * mI12 + L12 * mI22);
    const double A21 = -(L21 * mI11 + L22 * mI12), A22 = -(L21 * mI12 + L22 * mI22);
    const double B11 = -(U11 * pI11 + U12 * pI12), B12 = -(U11 * pI12 + U12 * pI22);
    const double B21 = -(U21 * pI11 + U22 * pI12), B22 = -(U21 * pI12 + U22 * pI22);
    // D' = D + alpha U[lane-s] + beta L[lane+s]   (symmetric; keep one off-diagonal)
    D11 += A11 * mU11 + A12 * mU21 + B11 * pL11 + B12 * pL21;
    D12 += A11 * mU12 + A12 * mU22 + B11 * pL12 + B12 * pL22;
    D22 += A21 * mU12 + A22 * mU22 + B21 * pL12 + B22 * pL22;
    F1 += A11 * mF1 + A12 * mF2 + B11 * pF1 + B12 * pF2;
    F2 += A21 * mF1 + A22 * mF2 + B21 * pF1 + B22 * pF2;
    // L' = alpha L[lane-s],  U' = beta U[lane+s]
    const double nL11 = A11 * mL11 + A12 * mL21, nL12 = A11 * mL12 + A12 * mL22;
    const double nL21 = A21 * mL11 + A22 * mL21, nL22 = A21 * mL12 + A22 * mL22;
    const double nU11 = B11 * pU11 + B12 * pU21, nU12 = B11 * pU12 + B12 * pU22;
    const double nU21 = B21 * pU11 + B22 * pU21, nU22 = B21 * pU12 + B22 * pU22;
    L11 = nL11; L12 = nL12; L21 = nL21; L22 = nL22;
    U11 = nU11; U12 = nU12; U21 = nU21; U22 = nU22;
  }
  double u1, u2;
  {
    const double det = D11 * D22 - D12 * D12;
    bad |= (D11 <= 0.0) | (det <= 0.0);
    const double idet = 1.0 / det;
    u1 = (D22 * F1 - D12 * F2) * idet;
    u2 = (D11 * F2 - D12 * F1) * idet;
  }
  double ul1 = shfl_up_d(u1, 1), ul2 = shfl_up_d(u2, 1);
  if (lane == 0) ul1 = ul2 = 0.0;

  // ---- 4. back-substitution of the interior ----
  {
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;  // q1=l1[i-1], q2=l2[i-1], q3=l2[i-2]
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double ba = (i == 0) ? ca : 0.0;
      const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
      const double va = ba - q1 * a1 - q3 * a2;
      const double vb = bb - q1 * b1 - q3 * b2;
      rhs[i] -= va * ul1 + vb * ul2;
      a2 = a1; a1 = va;
      b2 = b1; b1 = vb;
      q3 = q2; q2 = P2[i]; q1 = P1[i];
    }
  }
  {
    double x1 = u1, x2 = u2;
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
      const double xi = rhs[i] * Pd[i] - P1[i] * x1 - P2[i] * x2;
      rhs[i] = xi;
      x2 = x1;
      x1 = xi;
    }
    rhs[n] = u1;
    rhs[n + 1] = u2;
  }

  // ---- status: a non-positive pivot anywhere means the matrix is not positive definite; the
  // reference reports the first failing pivot of the natural-order factorisation ----
  int status = 0;
  const bool any_bad = __ballot(bad) != 0ull;
  if (any_bad && sys_valid) {
    if (lane == 0) {
      const SysView<TIN, BWD> view = make_view<TIN, BWD>(p, ws, b, d, T);
      status = first_bad_pivot<2, TIN, BWD>(view, ws);
    }
    status = __shfl(status, 0);
  }
  if (sys_valid && lane == 0 && p.status) p.status[(size_t)b * sd + d] = status;
  const bool zero_out = status != 0;

  // ---- 5. output ----
  if (!BWD) {
    TOUT *tileO = (TOUT *)tileA_raw;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int t = f0 + i;
      if (t < T) tileO[wv * TP + tile_idx<M>(t)] = zero_out ? (TOUT)0 : (TOUT)rhs[i];
    }
    __syncthreads();
    store_tile<M, TOUT>(tileO, (TOUT *)p.out + (size_t)b * Tmax * sd + d0, sd, T, Tmax, gvalid, tid);
  } else {
    // grad[t, w*sd+d] = tau_w[t] * (cm x[t-1] + c0 x[t] + cp x[t+1])      (paramgen/_mlpg.py:202-281)
    double xl = shfl_up_d(rhs[M - 1], 1), xr = shfl_dn_d(rhs[0], 1);
    if (lane == 0) xl = 0.0;
    if (lane == 63) xr = 0.0;
    TOUT *tileO = (TOUT *)tileB_raw;
    for (int w = 0; w < nw; ++w) {
      const int l = ws.l[w], u = ws.u[w];
      const double *cw = ws.c + ws.off[w];
      const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
      __syncthreads();  // previous store_tile / tile users done
      if (var_mode == MLPG_HIP_VAR_FRAME) load_tile<M, TIN>(tileV, var_b + w * sd + d0, D, T, gvalid, tid);
      __syncthreads();
      double tau_glob = 1.0;
      if (var_mode == MLPG_HIP_VAR_GLOBAL && sys_valid) tau_glob = recip_in_dtype<TIN>(var_b[w * sd + d]);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const int t = f0 + i;
        if (t < T) {
          double tau = 0.0;
          if (!(w != 0 && (mw == 0 || t < mw || t >= T - mw)))
            tau = (var_mode == MLPG_HIP_VAR_FRAME) ? recip_in_dtype<TIN>(tileV[wv * TP + tile_idx<M>(t)]) : tau_glob;
          const double xm = (i == 0) ? xl : rhs[i > 0 ? i - 1 : 0];
          const double xp = (i == M - 1) ? xr : rhs[i < M - 1 ? i + 1 : M - 1];
          const double xpv = (t + 1 < T) ? xp : 0.0;
          const double gval = tau * (cm * xm + c0 * rhs[i] + cp * xpv);
          tileO[wv * TP + tile_idx<M>(t)] = zero_out ? (TOUT)0 : (TOUT)gval;
        }
      }
      __syncthreads();
      store_tile<M, TOUT>(tileO, (TOUT *)p.out + (size_t)b * Tmax * D + w * sd + d0, D, T, Tmax, gvalid, tid);
    }
  }
}

template <int M, typename TIN, typename TOUT, bool BWD>
int launch_m(hipStream_t st, const Problem &p, const WinSet &ws) {
  constexpr int G = Geo<M>::G;
  const int ngrp = (p.sd + G - 1) / G;
  const int nslots = ((p.B + 7) / 8) * ngrp;
  const size_t lds = sizeof(double) * 2 * G * Geo<M>::TPAD;
  auto kern = wave_kernel<M, TIN, TOUT, BWD>;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nslots * 8), dim3(Geo<M>::NT), lds, st, p, ws, ngrp, nslots);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws) {
  if (p.Tmax <= 64 * 4) return launch_m<4, TIN, TOUT, BWD>(st, p, ws);
  if (p.Tmax <= 64 * 8) return launch_m<8, TIN, TOUT, BWD>(st, p, ws);
  if (p.Tmax <= 64 * 16) return launch_m<16, TIN, TOUT, BWD>(st, p, ws);
  return launch_m<32, TIN, TOUT, BWD>(st, p, ws);
}

}  // namespace

bool wave_supported(const Problem &p, const WinSet &ws) {
  if (p.Tmax > 64 * 32 || p.Tmax < 1) return false;
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return true;
}

int launch_wave(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws,
                int device) {
  (void)device;
  if (!backward) {
    return dtype == MLPG_HIP_F32 ? launch_t<float, float, false>(st, p, ws)
                                 : launch_t<double, double, false>(st, p, ws);
  }
  if (dtype == MLPG_HIP_F32)
    return out_dtype == MLPG_HIP_F32 ? launch_t<float, float, true>(st, p, ws)
                                     : launch_t<float, double, true>(st, p, ws);
  return out_dtype == MLPG_HIP_F32 ? launch_t<double, float, true>(st, p, ws)
                                   : launch_t<double, double, true>(st, p, ws);
}

}  // namespace mlpg
