// Wave-per-system MLPG kernels (the fast path; algo = MLPG_HIP_ALGO_WAVE / AUTO).
//
// One 64-lane wavefront solves one (utterance, static dim) system; a workgroup of G = 4
// wavefronts takes 4 consecutive static dims of one utterance, so that the (T, D) row-major
// inputs are staged through LDS in runs of 4 columns per frame.  Two workgroups share a CU
// (<= 256 VGPRs, ~70 KB LDS each): while one waits for its tiles the other computes.  Windows
// must have extents l, u <= 1 (half-bandwidth <= 2): the static/delta/delta-delta family;
// anything else goes to the generic kernel (mlpg_generic.hip).
//
// Per system (T frames, lane p owns the chunk of M consecutive frames [pM, pM+M), 64*M >= T,
// frames >= T are identity rows):
//   1. Assembly.  Per window, the G systems' variance and mean columns are brought into two LDS
//      tiles -- by LDS-DMA (global_load_lds_dwordx4: asynchronous, no staging registers) when
//      the column runs are 16-byte aligned, else through registers -- and every lane
//      accumulates the pentadiagonal P (diagonal + 2 sub-diagonals) and b of its chunk in
//      REGISTERS.  In DMA mode the chunk's raw values are first copied to registers so that the
//      next window's transfer can start while this window's arithmetic runs.
//   2. Substructuring: each lane eliminates its M-2 interior frames (LDL^T, sequential) carrying
//      the two "left spike" columns that couple it to the previous lane's last 2 frames; the
//      elimination runs on into the lane's own last 2 frames (its separator), which yields the
//      Schur complement: a block-tridiagonal SPD system with 2x2 blocks over the 64 separators.
//   3. That reduced system is solved across the lanes by parallel cyclic reduction (6 steps,
//      cross-lane shuffles); the multipliers wait in LDS meanwhile ("parking") instead of
//      being spilled to scratch.
//   4. Each lane back-substitutes its interior (two short sweeps).
//   5. The trajectory goes back through LDS and is stored in runs of G columns.
// The factor never leaves the chip; HBM traffic is the algorithmic minimum (means + variances
// read once, trajectory written once).
//
// Reference semantics reproduced (paramgen/_mlpg.py:92-199): tau = 1/var (float32 inputs: in
// float32, as _mlpg.py:188), dynamic-window precisions zeroed on the first/last mw frames,
// float64 arithmetic, output cast to the input dtype, status = index of the first non-positive
// pivot of the NATURAL-order Cholesky (linalg.pyx:79-82; found by a sequential re-scan on the
// rare failing system).  Backward (paramgen/_mlpg.py:202-281): same factor, right-hand side =
// grad_out, epilogue grad[t, w*sd+d] = tau_w[t] * (W_w x)[t].
#pragma once
#include "assemble.h"

#ifndef MLPG_WAVE_DMA
#define MLPG_WAVE_DMA 1  // 0: never use the LDS-DMA staging path (register staging only)
#endif
#ifndef MLPG_WAVE_DMA_AUX
#define MLPG_WAVE_DMA_AUX 0  // cache policy bits of the LDS-DMA loads (1 sc0, 2 nt, 16 sc1)
#endif
#ifndef MLPG_WAVE_PARK
#define MLPG_WAVE_PARK 1  // park the multipliers in LDS during the cyclic reduction
#endif
#ifndef MLPG_WAVE_ABLATE
#define MLPG_WAVE_ABLATE 0  // profiling only: 1 skip the solve, 2 skip the global loads, 4 = 1 + skip assembly math
#endif

#ifdef MLPG_WAVE_TIMING
#define MLPG_TICK(k)                                                   \
  do {                                                                 \
    const long long t_now_ = (long long)__builtin_readcyclecounter(); \
    tq[k] += t_now_ - t_prev;                                          \
    t_prev = t_now_;                                                   \
  } while (0)
#else
#define MLPG_TICK(k) do {} while (0)
#endif

namespace mlpg {
namespace {

constexpr int kG = 4;     // systems (wavefronts) per workgroup
constexpr int kSkew = 1;  // padding slot per chunk in the register-staged tile layout

// 1/d to ~1 ulp: hardware seed + two Newton steps (an IEEE-exact f64 divide is ~2x the
// instructions; the difference, 1e-16 relative, is far below every tolerance on this path).
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}

// tau = 1/var: float32 inputs keep the reference's float32 reciprocal (_mlpg.py:188)
template <typename T>
__device__ __forceinline__ double tau_of(T v);
template <>
__device__ __forceinline__ double tau_of<float>(float v) {
  return (double)__fdiv_rn(1.0f, v);
}
template <>
__device__ __forceinline__ double tau_of<double>(double v) {
  return fast_rcp(v);
}

constexpr int log2i(int m) { return m == 4 ? 2 : m == 8 ? 3 : m == 16 ? 4 : 5; }

// ---- tile layouts ------------------------------------------------------------------------
// Both hold G systems x (64*M) frames of one feature column group; idx(t, g) is the element
// offset of frame t of system g.  Within a lane's chunk, consecutive frames are ESTRIDE apart.
//
// Register-staged layout: [g][t + t/M]; the one-slot skew per chunk makes the solver lanes'
// stride M+1 (odd) elements: conflict-free ds_read_b64.
template <int M>
struct RegLayout {
  static constexpr int TPAD = 64 * (M + kSkew) + 2;  // pitch per system
  static constexpr int ESTRIDE = 1;
  static constexpr int ELEMS = kG * TPAD;
  static __device__ __forceinline__ int idx(int t, int g) { return g * TPAD + t + (t >> log2i(M)) * kSkew; }
};
// LDS-DMA layout.  One global_load_lds_dwordx4 moves 64 lanes x 16 bytes and the LDS
// destination is wave-uniform base + lane * 16, so the image is dictated by the lane ->
// (frame, column) assignment: a block = FB consecutive frames x G columns (each frame's G
// columns are one contiguous run in HBM), padded by 16 bytes so that the solver lanes (stride M
// frames) spread over the banks (2-way conflicts at worst).  FB is a multiple of M, so a chunk
// never straddles a block.
template <int M, typename TIN>
struct DmaLayout {
  static constexpr int EPL = 16 / (int)sizeof(TIN);  // elements per lane per transfer
  static constexpr int LPF = kG / EPL;               // lanes per frame
  static constexpr int FB = 64 / LPF;                // frames per block (= per instruction)
  static constexpr int NB = 64 * M / FB;             // blocks per tile
  static constexpr int BS = FB * kG + EPL;           // block stride in elements (16 B pad)
  static constexpr int ESTRIDE = kG;
  static constexpr int ELEMS = NB * BS;
  static_assert(FB % M == 0, "a chunk must not straddle a DMA block");
  static __device__ __forceinline__ int idx(int t, int g) { return (t / FB) * BS + (t % FB) * kG + g; }
};

template <int M, typename TIN>
constexpr int tile_bytes() {
  constexpr int a = RegLayout<M>::ELEMS * 8;  // also the output staging / parking footprint
  constexpr int b = DmaLayout<M, TIN>::ELEMS * (int)sizeof(TIN);
  return ((a > b ? a : b) + 15) / 16 * 16;
}

// ---- staging: global -> LDS ----------------------------------------------------------------
// Register-staged: every thread owns exactly M elements of the tile (t_k = tid / G + 64 k,
// g = tid % G): consecutive threads read consecutive columns of one frame.
template <int M, typename TIN>
__device__ __forceinline__ void load_tile_regs(TIN *__restrict__ tile, const TIN *__restrict__ src, int row_stride,
                                               int T, int gvalid, int tid) {
  const int g = tid % kG, t0 = tid / kG;
  TIN v[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const int t = t0 + 64 * k;
    v[k] = (t < T && g < gvalid) ? src[(size_t)t * row_stride + g] : (TIN)0;
  }
  TIN *dst = tile + RegLayout<M>::idx(t0, g);
  constexpr int kStep = 64 + (64 >> log2i(M)) * kSkew;  // idx(t + 64) - idx(t)
#pragma unroll
  for (int k = 0; k < M; ++k) dst[k * kStep] = v[k];
}

// LDS-DMA: wavefront wv issues blocks wv, wv + G, ...  Needs 16-byte aligned column runs.
template <int M, typename TIN>
__device__ __forceinline__ void load_tile_dma(TIN *__restrict__ tile, const TIN *__restrict__ src, int row_stride,
                                              int T, int gvalid, int wv, int lane) {
  using L = DmaLayout<M, TIN>;
  const int fl = lane / L::LPF, j = lane % L::LPF;
  const bool cols_ok = (j + 1) * L::EPL <= gvalid;
  const TIN *ptr = src + (size_t)(wv * L::FB + fl) * row_stride + j * L::EPL;
  const size_t step = (size_t)kG * L::FB * row_stride;
#pragma unroll
  for (int k = 0; k < (L::NB + kG - 1) / kG; ++k) {
    const int q = wv + k * kG;
    if (q >= L::NB || q * L::FB >= T) break;  // wave-uniform
    if (q * L::FB + fl < T && cols_ok) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ptr + k * step),
                                       (__attribute__((address_space(3))) void *)(tile + q * L::BS), 16, 0, MLPG_WAVE_DMA_AUX);
    }
  }
}

// LDS -> global: dst[t * row_stride + g] = tile[g][t] (register layout) for t < T, 0 for T <= t < Tmax.
// Each thread moves a PAIR of adjacent columns per frame (one 16-byte store for float64 when the
// run is 16-byte aligned), 128 frames apart per iteration; all index arithmetic is incremental.
template <int M, typename TOUT>
__device__ __forceinline__ void store_tile(const TOUT *__restrict__ tile, TOUT *__restrict__ dst, int row_stride, int T,
                                           int Tmax, int gvalid, int tid, bool aligned_pairs) {
  using RL = RegLayout<M>;
  const int g0 = (tid & 1) * 2, t0 = tid >> 1;
  if (g0 >= gvalid) return;
  const bool two = g0 + 1 < gvalid;
  constexpr int kStepL = 128 + (128 >> log2i(M)) * kSkew;  // idx(t + 128) - idx(t)
  int li = RL::idx(t0, g0);
  TOUT *gp = dst + (size_t)t0 * row_stride + g0;
  const size_t gstep = (size_t)128 * row_stride;
  struct alignas(2 * sizeof(TOUT)) Pair { TOUT a, b; };
  for (int t = t0; t < Tmax; t += 128, li += kStepL, gp += gstep) {
    const bool live = t < T;
    const TOUT va = live ? tile[li] : (TOUT)0;
    const TOUT vb = (live && two) ? tile[li + RL::TPAD] : (TOUT)0;
    if (two && aligned_pairs) {
      Pair pr;
      pr.a = va;
      pr.b = vb;
      *reinterpret_cast<Pair *>(gp) = pr;
    } else {
      gp[0] = va;
      if (two) gp[1] = vb;
    }
  }
}

// ---- steps 2-4 for one system held in registers --------------------------------------------
// Interior elimination with left spikes, block-tridiagonal reduced system by parallel cyclic
// reduction across the lanes, interior back-substitution.  On return rhs[] holds the solution
// of the chunk.  Returns true if a non-positive pivot was met anywhere (then the matrix is not
// positive definite).  parkA/parkB: per-lane LDS scratch of M-2 doubles each (PARK only).
// What a second solve with the SAME matrix needs from the cyclic reduction: per step the inverse of the diagonal block
// and the coupling block as they stood when the step began, and the last step's inverse.  (The interior multipliers
// stay in Pd/P1/P2.)  46 doubles per lane.
struct PcrKeep {
  double I11[6], I12[6], I22[6], L11[6], L12[6], L21[6], L22[6];
  double F11, F12, F22, Fi;  // the last block (D22, D12, D11) and 1 / det
};

template <int M, bool PARK, bool KEEP = false>
__device__ __forceinline__ bool solve_chunk(double (&Pd)[M], double (&P1)[M], double (&P2)[M], double (&rhs)[M],
                                            int lane, double *parkA, double *parkB, PcrKeep *keep = nullptr) {
  constexpr int n = M - 2;  // interior frames per lane; frames n, n+1 form the lane's separator
  // coupling of this chunk's first two frames to the previous lane's separator (frames f0-2, f0-1)
  double ca = __shfl_up(P2[M - 2], 1);  // P[f0,   f0-2]
  double cb = __shfl_up(P1[M - 1], 1);  // P[f0,   f0-1]
  double cc = __shfl_up(P2[M - 1], 1);  // P[f0+1, f0-1]
  if (lane == 0) ca = cb = cc = 0.0;

  bool bad = false;
  double t00 = 0.0, t01 = 0.0, t11 = 0.0, h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    const double dd = Pd[i];
    bad |= (dd <= 0.0);
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
    const double n00 = __shfl_down(t00, 1), n01 = __shfl_down(t01, 1), n11 = __shfl_down(t11, 1);
    const double nh0 = __shfl_down(h0, 1), nh1 = __shfl_down(h1, 1);
    if (lane < 63) {
      D11 -= n00; D12 -= n01; D22 -= n11;
      F1 -= nh0; F2 -= nh1;
    }
  }
  if (lane == 0) L11 = L12 = L21 = L22 = 0.0;

  // The multipliers are not needed again until the back-substitution: with PARK they wait in LDS
  // instead of being spilled to scratch under the register pressure of the cyclic reduction.
  if (PARK) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      parkA[i] = P1[i];
      parkB[i] = P2[i];
    }
  }

  // ---- block-tridiagonal reduced system over the 64 separators: parallel cyclic reduction ----
  // Row j at stride s:  L_j u_{j-s} + D_j u_j + L_{j+s}^T u_{j+s} = F_j   (the matrix is symmetric:
  // only the sub-diagonal blocks L are carried; the super-diagonal block of row j is L_{j+s}^T).
  //   D_j' = D_j - L_j D_{j-s}^-1 L_j^T - L_{j+s}^T D_{j+s}^-1 L_{j+s}
  //   F_j' = F_j - L_j D_{j-s}^-1 F_{j-s} - L_{j+s}^T D_{j+s}^-1 F_{j+s}
  //   L_j' = -L_j D_{j-s}^-1 L_{j-s}
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const double det = D11 * D22 - D12 * D12;
    bad |= (D11 <= 0.0) | (det <= 0.0);
    const double idet = fast_rcp(det);
    const double I11 = D22 * idet, I12 = -D12 * idet, I22 = D11 * idet;
    if (KEEP) {
      constexpr int kLog[33] = {0, 0, 1, 0, 2, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5};
      const int k = kLog[s];
      keep->I11[k] = I11; keep->I12[k] = I12; keep->I22[k] = I22;
      keep->L11[k] = L11; keep->L12[k] = L12; keep->L21[k] = L21; keep->L22[k] = L22;
    }
    // G_j = D_j^-1 F_j and H_j = D_j^-1 L_j are what the neighbours need from row j
    const double G1 = I11 * F1 + I12 * F2, G2 = I12 * F1 + I22 * F2;
    const double H11 = I11 * L11 + I12 * L21, H12 = I11 * L12 + I12 * L22;
    const double H21 = I12 * L11 + I22 * L21, H22 = I12 * L12 + I22 * L22;
    const bool hasm = lane - s >= 0, hasp = lane + s < 64;
    // The neighbour data is fetched and consumed in SMALL groups separated by scheduling
    // barriers: otherwise the scheduler hoists all 38 ds_bpermute results to the top of the step
    // and the live set (on top of the 4M persistent doubles) spills.
    double nL11, nL12, nL21, nL22;
    {  // row j-s: inverse block -> K = L_j D_{j-s}^-1;  D -= K L_j^T
      double mI11 = __shfl_up(I11, s), mI12 = __shfl_up(I12, s), mI22 = __shfl_up(I22, s);
      if (!hasm) { mI11 = mI12 = mI22 = 0.0; }
      const double K11 = L11 * mI11 + L12 * mI12, K12 = L11 * mI12 + L12 * mI22;
      const double K21 = L21 * mI11 + L22 * mI12, K22 = L21 * mI12 + L22 * mI22;
      D11 -= K11 * L11 + K12 * L12;
      D12 -= K11 * L21 + K12 * L22;
      D22 -= K21 * L21 + K22 * L22;
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // row j-s: G -> F -= L_j G_{j-s}
      double mG1 = __shfl_up(G1, s), mG2 = __shfl_up(G2, s);
      if (!hasm) { mG1 = mG2 = 0.0; }
      F1 -= L11 * mG1 + L12 * mG2;
      F2 -= L21 * mG1 + L22 * mG2;
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // row j-s: H -> L' = -L_j H_{j-s}
      double mH11 = __shfl_up(H11, s), mH12 = __shfl_up(H12, s), mH21 = __shfl_up(H21, s), mH22 = __shfl_up(H22, s);
      if (!hasm) { mH11 = mH12 = mH21 = mH22 = 0.0; }
      nL11 = -(L11 * mH11 + L12 * mH21); nL12 = -(L11 * mH12 + L12 * mH22);
      nL21 = -(L21 * mH11 + L22 * mH21); nL22 = -(L21 * mH12 + L22 * mH22);
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // row j+s: its coupling block L_{j+s} and H_{j+s}, G_{j+s}:  D -= L^T H;  F -= L^T G
      double pL11 = __shfl_down(L11, s), pL12 = __shfl_down(L12, s), pL21 = __shfl_down(L21, s), pL22 = __shfl_down(L22, s);
      if (!hasp) { pL11 = pL12 = pL21 = pL22 = 0.0; }
      {
        double pG1 = __shfl_down(G1, s), pG2 = __shfl_down(G2, s);
        if (!hasp) { pG1 = pG2 = 0.0; }
        F1 -= pL11 * pG1 + pL21 * pG2;
        F2 -= pL12 * pG1 + pL22 * pG2;
      }
      __builtin_amdgcn_sched_barrier(0);
      double pH11 = __shfl_down(H11, s), pH12 = __shfl_down(H12, s), pH21 = __shfl_down(H21, s), pH22 = __shfl_down(H22, s);
      if (!hasp) { pH11 = pH12 = pH21 = pH22 = 0.0; }
      D11 -= pL11 * pH11 + pL21 * pH21;
      D12 -= pL11 * pH12 + pL21 * pH22;
      D22 -= pL12 * pH12 + pL22 * pH22;
    }
    L11 = nL11; L12 = nL12; L21 = nL21; L22 = nL22;
    __builtin_amdgcn_sched_barrier(0);
  }
  double u1, u2;
  {
    const double det = D11 * D22 - D12 * D12;
    bad |= (D11 <= 0.0) | (det <= 0.0);
    const double idet = fast_rcp(det);
    u1 = (D22 * F1 - D12 * F2) * idet;
    u2 = (D11 * F2 - D12 * F1) * idet;
    if (KEEP) { keep->F11 = D22; keep->F12 = D12; keep->F22 = D11; keep->Fi = idet; }
  }
  double ul1 = __shfl_up(u1, 1), ul2 = __shfl_up(u2, 1);
  if (lane == 0) ul1 = ul2 = 0.0;

  if (PARK) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      P1[i] = parkA[i];
      P2[i] = parkB[i];
    }
  }
  // ---- back-substitution of the interior: z = g - (left spikes) u_left, then L^T x = D^-1 z ----
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
  return bad;
}

// ---- a further right-hand side for the matrix solve_chunk<M, false, true> has just factorised ---------------
// Pd/P1/P2[0..M-2) hold 1/d, l1, l2 of the interior, Pd/P1/P2's separator couplings are untouched, `keep` holds the
// cyclic reduction's blocks.  Every operation on the right-hand side is the one solve_chunk performs, in its order
// (so the result equals a full second solve bit for bit) -- none of the matrix arithmetic, a fifth of the cross-lane
// traffic.  On return rhs[] holds the solution.
template <int M>
__device__ __forceinline__ void solve_again(const double (&Pd)[M], const double (&P1)[M], const double (&P2)[M],
                                            double (&rhs)[M], int lane, const PcrKeep &keep) {
  constexpr int n = M - 2;
  double ca = __shfl_up(P2[M - 2], 1);
  double cb = __shfl_up(P1[M - 1], 1);
  double cc = __shfl_up(P2[M - 1], 1);
  if (lane == 0) ca = cb = cc = 0.0;
  double h0 = 0.0, h1 = 0.0;
  double g1 = 0.0, g2 = 0.0, va1 = 0.0, va2 = 0.0, vb1 = 0.0, vb2 = 0.0;
  double l1p = 0.0, l2p = 0.0, l2pp = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    const double dinv = Pd[i], l1 = P1[i], l2 = P2[i];
    const double gi = rhs[i] - l1p * g1 - l2pp * g2;
    const double ba = (i == 0) ? ca : 0.0;
    const double bb = (i == 0) ? cb : ((i == 1) ? cc : 0.0);
    const double va = ba - l1p * va1 - l2pp * va2;
    const double vb = bb - l1p * vb1 - l2pp * vb2;
    const double wa = va * dinv, wb = vb * dinv;
    h0 += wa * gi;
    h1 += wb * gi;
    rhs[i] = gi;
    g2 = g1; g1 = gi;
    va2 = va1; va1 = va;
    vb2 = vb1; vb1 = vb;
    l2pp = l2p; l2p = l2; l1p = l1;
  }
  rhs[n] -= l1p * g1 + l2pp * g2;
  rhs[n + 1] -= l2p * g1;
  double F1 = rhs[n], F2 = rhs[n + 1];
  {
    const double nh0 = __shfl_down(h0, 1), nh1 = __shfl_down(h1, 1);
    if (lane < 63) { F1 -= nh0; F2 -= nh1; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int s = 1 << k;
    const double G1 = keep.I11[k] * F1 + keep.I12[k] * F2, G2 = keep.I12[k] * F1 + keep.I22[k] * F2;
    const bool hasm = lane - s >= 0, hasp = lane + s < 64;
    {
      double mG1 = __shfl_up(G1, s), mG2 = __shfl_up(G2, s);
      if (!hasm) { mG1 = mG2 = 0.0; }
      F1 -= keep.L11[k] * mG1 + keep.L12[k] * mG2;
      F2 -= keep.L21[k] * mG1 + keep.L22[k] * mG2;
    }
    {
      double pL11 = __shfl_down(keep.L11[k], s), pL12 = __shfl_down(keep.L12[k], s), pL21 = __shfl_down(keep.L21[k], s), pL22 = __shfl_down(keep.L22[k], s);
      double pG1 = __shfl_down(G1, s), pG2 = __shfl_down(G2, s);
      if (!hasp) { pL11 = pL12 = pL21 = pL22 = 0.0; pG1 = pG2 = 0.0; }
      F1 -= pL11 * pG1 + pL21 * pG2;
      F2 -= pL12 * pG1 + pL22 * pG2;
    }
  }
  const double u1 = (keep.F11 * F1 - keep.F12 * F2) * keep.Fi;
  const double u2 = (keep.F22 * F2 - keep.F12 * F1) * keep.Fi;
  double ul1 = __shfl_up(u1, 1), ul2 = __shfl_up(u2, 1);
  if (lane == 0) ul1 = ul2 = 0.0;
  {
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
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
}

// ---- the kernel ------------------------------------------------------------------------------
// VM = variance mode (MLPG_HIP_VAR_*), DMA = LDS-DMA staging, MINW = waves per SIMD to fit
// (2: two workgroups per CU, <= 256 VGPRs; 1: one workgroup per CU, 512 registers).
template <int M, int MINW, typename TIN, typename TOUT, bool BWD, bool DMA, int VM>
__global__ __launch_bounds__(kG * 64, MINW) void wave_kernel(Problem p, WinSet ws, int ngrp, int nslots) {
  using RL = RegLayout<M>;
  using DL = DmaLayout<M, TIN>;
  constexpr int kTileBytes = tile_bytes<M, TIN>();
  constexpr bool kVarTile = VM == MLPG_HIP_VAR_FRAME;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char *tileA = smem, *tileB = smem + kTileBytes;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // XCD-aware decode: workgroup w runs on XCD w % 8; consecutive slots of one XCD are the
  // static-dim groups of one utterance, so sibling groups (which share 128-byte lines of the
  // row-major input) hit the same L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= nslots) return;
  const int b = (slot / ngrp) * 8 + xcd, dgrp = slot % ngrp;
  if (b >= p.B) return;
  const int sd = p.sd, Tmax = p.Tmax;
  const int wp = (!BWD && p.pitch) ? p.pitch : sd;  // columns between a dim's windows (a piece of a stream: > sd)
  const int ldi = (int)p.ld_in, ldg = (int)p.ld_gout, ldo = (int)p.ld_out;  // row strides (elements)
  const int d0 = dgrp * kG, d = d0 + wv;
  const int gvalid = sd - d0 < kG ? sd - d0 : kG;
  const bool sys_valid = d < sd;
  int T = p.lengths ? p.lengths[b] : Tmax;
  T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
  const int mw = ws.mw, nw = ws.nw;
  // paired output stores need every run start (b, t, [w], d0) to be 2-element aligned
  const bool out_pairs_ok = (ldo % 2 == 0) && (!BWD || sd % 2 == 0) && (((uintptr_t)p.out & (2 * sizeof(TOUT) - 1)) == 0);

  const TIN *mean_b = BWD ? nullptr : (const TIN *)p.mean + (size_t)b * Tmax * ldi;
  const TIN *var_b = (const TIN *)p.var;
  if (kVarTile) var_b += (size_t)b * Tmax * ldi;
  const TIN *gout_b = BWD ? (const TIN *)p.grad_out + (size_t)b * Tmax * ldg : nullptr;

  const int f0 = lane * M;  // first frame of this lane's chunk
#ifdef MLPG_WAVE_TIMING
  long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // setup, wait-for-tiles, lds->regs, dma issue, assembly, solve, status, output
  long long t_prev = (long long)__builtin_readcyclecounter();
#endif
  // liveness of the chunk's frames plus one halo frame on each side, bit i+1 <-> frame f0+i:
  // static window: 0 <= t < T;  dynamic windows: mw <= t < T - mw (and nothing at all if mw == 0,
  // Python's precisions[-0:] = 0 slice, _mlpg.py:191-193)
  unsigned long long liveS = 0ull, liveD = 0ull;  // M + 2 <= 34 bits
#pragma unroll
  for (int i = -1; i <= M; ++i) {
    const int t = f0 + i;
    if (t >= 0 && t < T) liveS |= 1ull << (i + 1);
    if (mw != 0 && t >= mw && t < T - mw) liveD |= 1ull << (i + 1);
  }

  // per-lane element offsets into the tiles: chunk frame i sits at base + i * ESTRIDE
  const int baseR = RL::idx(f0, wv);
  const int loR = RL::idx(f0 > 0 ? f0 - 1 : 0, wv);
  const int hiR = RL::idx(f0 + M < 64 * M ? f0 + M : 64 * M - 1, wv);
  const int baseD = DL::idx(f0, wv);
  const int loD = DL::idx(f0 > 0 ? f0 - 1 : 0, wv);
  const int hiD = DL::idx(f0 + M < 64 * M ? f0 + M : 64 * M - 1, wv);

  // ---- 1. assembly: Pd[i] = P[f,f], P1[i] = P[f+1,f], P2[i] = P[f+2,f], rhs[i], f = f0 + i ----
  double Pd[M], P1[M], P2[M], rhs[M];
#pragma unroll
  for (int i = 0; i < M; ++i) Pd[i] = P1[i] = P2[i] = rhs[i] = 0.0;

  TIN *tileV = (TIN *)tileA, *tileM = (TIN *)tileB;

  if (BWD) {
    // right-hand side = grad_out[:, d]; runs of G columns with row stride sd
    if (DMA) {
      load_tile_dma<M, TIN>(tileM, gout_b + d0, ldg, T, gvalid, wv, lane);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < M; ++i) rhs[i] = ((liveS >> (i + 1)) & 1ull) ? (double)tileM[baseD + i * DL::ESTRIDE] : 0.0;
    } else {
      load_tile_regs<M, TIN>(tileM, gout_b + d0, ldg, T, gvalid, tid);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < M; ++i) rhs[i] = ((liveS >> (i + 1)) & 1ull) ? (double)tileM[baseR + i] : 0.0;
    }
    __syncthreads();
  }

  MLPG_TICK(0);
  if (DMA && MLPG_WAVE_ABLATE != 2) {
    if (kVarTile) load_tile_dma<M, TIN>(tileV, var_b + d0, ldi, T, gvalid, wv, lane);
    if (!BWD) load_tile_dma<M, TIN>(tileM, mean_b + d0, ldi, T, gvalid, wv, lane);
  }
  MLPG_TICK(3);
  for (int w = 0; w < nw; ++w) {
    const int l = ws.l[w], u = ws.u[w];
    const double *cw = ws.c + ws.off[w];
    const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;  // W[t,t-1], W[t,t], W[t,t+1]
    double tau_glob = 1.0;
    if (VM == MLPG_HIP_VAR_GLOBAL && sys_valid) tau_glob = tau_of<TIN>(var_b[w * wp + d]);
    const double c00 = c0 * c0, cpp = cp * cp, cmm = cm * cm, cp0 = cp * c0, c0m = c0 * cm, cpm = cp * cm;
    const unsigned long long live = w ? liveD : liveS;

    // raw values of the chunk (+ halo) in registers
    TIN rv[M + 2], rm[M + 2];
    if (DMA) {
      __syncthreads();  // drains every wavefront's DMA queue: window w has landed
      MLPG_TICK(1);
      if (kVarTile) {
        rv[0] = tileV[loD];
        rv[M + 1] = tileV[hiD];
#pragma unroll
        for (int i = 0; i < M; ++i) rv[i + 1] = tileV[baseD + i * DL::ESTRIDE];
      }
      if (!BWD) {
        rm[0] = tileM[loD];
        rm[M + 1] = tileM[hiD];
#pragma unroll
        for (int i = 0; i < M; ++i) rm[i + 1] = tileM[baseD + i * DL::ESTRIDE];
      }
      __syncthreads();  // every wavefront holds its values: the tiles can be refilled now, and
                        // the transfer runs behind this window's arithmetic
      MLPG_TICK(2);
      if (w + 1 < nw && MLPG_WAVE_ABLATE != 2) {
        if (kVarTile) load_tile_dma<M, TIN>(tileV, var_b + (w + 1) * wp + d0, ldi, T, gvalid, wv, lane);
        if (!BWD) load_tile_dma<M, TIN>(tileM, mean_b + (w + 1) * wp + d0, ldi, T, gvalid, wv, lane);
      }
      MLPG_TICK(3);
    } else {
      if (MLPG_WAVE_ABLATE != 2) {
        if (kVarTile) load_tile_regs<M, TIN>(tileV, var_b + w * wp + d0, ldi, T, gvalid, tid);
        if (!BWD) load_tile_regs<M, TIN>(tileM, mean_b + w * wp + d0, ldi, T, gvalid, tid);
      }
      __syncthreads();
      if (kVarTile) {
        rv[0] = tileV[loR];
        rv[M + 1] = tileV[hiR];
#pragma unroll
        for (int i = 0; i < M; ++i) rv[i + 1] = tileV[baseR + i];
      }
      if (!BWD) {
        rm[0] = tileM[loR];
        rm[M + 1] = tileM[hiR];
#pragma unroll
        for (int i = 0; i < M; ++i) rm[i + 1] = tileM[baseR + i];
      }
      __syncthreads();
    }
    // one pass over the chunk plus a halo frame on each side: frame t feeds rows t-1, t, t+1
#pragma unroll
    for (int i = -1; i <= (MLPG_WAVE_ABLATE == 4 ? 0 : M); ++i) {
      const bool lv = (live >> (i + 1)) & 1ull;
      double tau = kVarTile ? tau_of<TIN>(rv[i + 1]) : tau_glob;
      tau = lv ? tau : 0.0;
      double tm = 0.0;
      if (!BWD) {
        tm = tau * (double)rm[i + 1];
        tm = lv ? tm : 0.0;  // dead slots may hold anything (never-written LDS)
      }
      if (i >= 0 && i < M) {  // row f = t
        Pd[i] += c00 * tau;
        P1[i] += cp0 * tau;  // P[f+1,f] gets W[f,f+1] W[f,f] tau[f]
        if (!BWD) rhs[i] += c0 * tm;
      }
      if (i + 1 >= 0 && i + 1 < M) {  // row f = t+1
        Pd[i + 1] += cpp * tau;
        if (!BWD) rhs[i + 1] += cp * tm;
      }
      if (i - 1 >= 0 && i - 1 < M) {  // row f = t-1
        Pd[i - 1] += cmm * tau;
        P1[i - 1] += c0m * tau;  // W[f+1,f+1] W[f+1,f] tau[f+1]
        P2[i - 1] += cpm * tau;  // W[f+1,f+2] W[f+1,f] tau[f+1]
        if (!BWD) rhs[i - 1] += cm * tm;
      }
    }
    MLPG_TICK(4);
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

  // ---- 2-4. solve ----
#if MLPG_WAVE_ABLATE != 1 && MLPG_WAVE_ABLATE != 4
  constexpr bool kPark = (M >= 16) && MLPG_WAVE_PARK;
  // parking rows live in the (now idle) tiles, register layout footprint
  const bool bad = solve_chunk<M, kPark>(Pd, P1, P2, rhs, lane, (double *)tileA + wv * RL::TPAD + lane * (M + kSkew),
                                         (double *)tileB + wv * RL::TPAD + lane * (M + kSkew));
#else
  const bool bad = false;
#endif

  MLPG_TICK(5);
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
  if (sys_valid && lane == 0 && p.status) p.status[(size_t)b * p.ld_status + d] = status;
  const bool zero_out = status != 0;

  MLPG_TICK(6);
  // ---- 5. output ----
  if (!BWD) {
    TOUT *tileO = (TOUT *)tileA;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < M; ++i)
      if ((liveS >> (i + 1)) & 1ull) tileO[baseR + i] = zero_out ? (TOUT)0 : (TOUT)rhs[i];
    __syncthreads();
    store_tile<M, TOUT>(tileO, (TOUT *)p.out + (size_t)b * Tmax * ldo + d0, ldo, T, Tmax, gvalid, tid, out_pairs_ok);
  } else {
    // grad[t, w*sd+d] = tau_w[t] * (cm x[t-1] + c0 x[t] + cp x[t+1])      (paramgen/_mlpg.py:202-281)
    double xl = __shfl_up(rhs[M - 1], 1), xr = __shfl_down(rhs[0], 1);
    if (lane == 0) xl = 0.0;
    if (lane == 63) xr = 0.0;
    TOUT *tileO = (TOUT *)tileB;
    for (int w = 0; w < nw; ++w) {
      const int l = ws.l[w], u = ws.u[w];
      const double *cw = ws.c + ws.off[w];
      const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
      const unsigned long long live = w ? liveD : liveS;
      __syncthreads();  // previous store_tile / tile users done
      if (kVarTile) {
        if (DMA) load_tile_dma<M, TIN>(tileV, var_b + w * sd + d0, ldi, T, gvalid, wv, lane);
        else load_tile_regs<M, TIN>(tileV, var_b + w * sd + d0, ldi, T, gvalid, tid);
      }
      __syncthreads();
      double tau_glob = 1.0;
      if (VM == MLPG_HIP_VAR_GLOBAL && sys_valid) tau_glob = tau_of<TIN>(var_b[w * sd + d]);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if ((liveS >> (i + 1)) & 1ull) {
          double tau = 0.0;
          if ((live >> (i + 1)) & 1ull) tau = kVarTile ? tau_of<TIN>(DMA ? tileV[baseD + i * DL::ESTRIDE] : tileV[baseR + i]) : tau_glob;
          const double xm = (i == 0) ? xl : rhs[i > 0 ? i - 1 : 0];
          const double xp = (i == M - 1) ? xr : rhs[i < M - 1 ? i + 1 : M - 1];
          const double gval = tau * (cm * xm + c0 * rhs[i] + cp * xp);  // x == 0 on rows >= T
          tileO[baseR + i] = zero_out ? (TOUT)0 : (TOUT)gval;
        }
      }
      __syncthreads();
      store_tile<M, TOUT>(tileO, (TOUT *)p.out + (size_t)b * Tmax * ldo + w * sd + d0, ldo, T, Tmax, gvalid, tid, out_pairs_ok);
    }
  }
#ifdef MLPG_WAVE_TIMING
  // profiling build only: per-phase cycle counts of wavefront 0 of the first 64 workgroups
  // overwrite the head of the status array (run bench.py --no-check with MLPG_DUMP_STATUS=1)
  MLPG_TICK(7);
  __syncthreads();
  if (tid == 0 && p.status && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0) {
    for (int k = 0; k < 8; ++k) p.status[(blockIdx.x >> 3) * 8 + k] = (int)tq[k];
  }
#endif
}

// ---- launchers ---------------------------------------------------------------------------------
template <int M, typename TIN, typename TOUT, bool BWD, bool DMA, int VM>
int launch_k(hipStream_t st, const Problem &p, const WinSet &ws) {
  constexpr int MINW = (M <= 16) ? 2 : 1;
  const int ngrp = (p.sd + kG - 1) / kG;
  const int nslots = ((p.B + 7) / 8) * ngrp;
  constexpr size_t lds = 2 * (size_t)tile_bytes<M, TIN>();
  auto kern = wave_kernel<M, MINW, TIN, TOUT, BWD, DMA, VM>;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  note_launch(kCountWave);
  hipLaunchKernelGGL(kern, dim3(nslots * 8), dim3(kG * 64), lds, st, p, ws, ngrp, nslots);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

// LDS-DMA moves 16-byte runs: every run of G columns must start 16-byte aligned.
template <typename TIN>
bool dma_ok(const Problem &p) {
  constexpr int epl = 16 / (int)sizeof(TIN);
  if (MLPG_WAVE_DMA == 0) return false;
  if (kG % epl || p.ld_in % epl || p.ld_gout % epl || p.sd % epl || p.pitch % epl) return false;
  auto al = [](const void *q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
  return al(p.mean) && (p.var_mode != MLPG_HIP_VAR_FRAME || al(p.var)) && al(p.grad_out);
}

template <int M, typename TIN, typename TOUT, bool BWD, bool DMA>
int launch_v(hipStream_t st, const Problem &p, const WinSet &ws) {
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: return launch_k<M, TIN, TOUT, BWD, DMA, MLPG_HIP_VAR_FRAME>(st, p, ws);
    case MLPG_HIP_VAR_GLOBAL: return launch_k<M, TIN, TOUT, BWD, DMA, MLPG_HIP_VAR_GLOBAL>(st, p, ws);
    default: return launch_k<M, TIN, TOUT, BWD, DMA, MLPG_HIP_VAR_UNIT>(st, p, ws);
  }
}

template <int M, typename TIN, typename TOUT, bool BWD>
int launch_m(hipStream_t st, const Problem &p, const WinSet &ws) {
  if (dma_ok<TIN>(p)) return launch_v<M, TIN, TOUT, BWD, true>(st, p, ws);
  return launch_v<M, TIN, TOUT, BWD, false>(st, p, ws);
}

template <typename TIN, typename TOUT, bool BWD>
int launch_t(hipStream_t st, const Problem &p, const WinSet &ws) {
  if (p.Tmax <= 64 * 4) return launch_m<4, TIN, TOUT, BWD>(st, p, ws);
  if (p.Tmax <= 64 * 8) return launch_m<8, TIN, TOUT, BWD>(st, p, ws);
  if (p.Tmax <= 64 * 16) return launch_m<16, TIN, TOUT, BWD>(st, p, ws);
  return launch_m<32, TIN, TOUT, BWD>(st, p, ws);
}

}  // namespace
}  // namespace mlpg
