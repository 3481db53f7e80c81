// Fused unit-variance MLPG training step (BASELINE config 3; the reference's perf/autograd_mlpg_perf.py loop,
// autograd/_impl/mlpg.py:70-172 around nn.MSELoss):
//
//     y    = MLPG(means)                       unit variances: y = P^-1 sum_w W_w^T mu_w
//     loss = mean((y - target)^2)
//     grad = d loss / d means                  = W_w P^-1 (2 (y - target) / N), window by window
//
// in ONE launch of the wave-per-system scheme (mlpg_wave_impl.h): a wavefront assembles and solves its system, turns
// the trajectory into the loss gradient in registers, solves the same matrix again (unit variances: P depends on T
// and the windows only, it is re-formed without a single load) and writes the three gradient columns.  The
// trajectory never travels through HBM between forward and backward, the two solves share one launch, and the loss
// is summed in a fixed order by the last workgroup to finish -- where the eager autograd path issues two solver
// launches and a dozen framework kernels (0.26 ms of host time per step; 0.10 ms as a captured graph).
#include "mlpg_wave_impl.h"

#ifndef MLPG_FUSED_KEEP
#define MLPG_FUSED_KEEP 1   // 0: the backward solve re-forms and re-factorises the matrix (round 3's first version)
#endif

namespace mlpg {
namespace {

struct FusedArgs {
  const void *target;   // (B, Tmax, sd), input dtype
  void *y_out;          // (B, Tmax, sd) or nullptr
  double scale;         // d loss / d y = scale * (y - target): 2 / N
  double inv_n;         // loss = inv_n * sum (y - target)^2
  double *partials;     // one per workgroup
  unsigned *counter;    // arrivals (zero at launch; left zero)
  double *loss;         // the scalar
};

template <int M, typename TIN, bool DMA>
__global__ __launch_bounds__(kG * 64, 2) void wave_fused_kernel(Problem p, WinSet ws, int ngrp, int nslots, FusedArgs fa) {
  using RL = RegLayout<M>;
  using DL = DmaLayout<M, TIN>;
  constexpr int kTileBytes = tile_bytes<M, TIN>();
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char *tileA = smem, *tileB = smem + kTileBytes;
  __shared__ double wsum[kG];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / ngrp) * 8 + xcd, dgrp = slot % ngrp;
  const bool active = slot < nslots && b < p.B;  // (inactive workgroups still take part in the loss reduction)
  double lsum = 0.0;
  if (active) {
    const int sd = p.sd, Tmax = p.Tmax;
    const int ldi = (int)p.ld_in, ldo = (int)p.ld_out;
    const int d0 = dgrp * kG, d = d0 + wv;
    const int gvalid = sd - d0 < kG ? sd - d0 : kG;
    const bool sys_valid = d < sd;
    int T = p.lengths ? p.lengths[b] : Tmax;
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    const int mw = ws.mw, nw = ws.nw;
    const bool grad_pairs_ok = (ldo % 2 == 0) && (sd % 2 == 0) && (((uintptr_t)p.out & (2 * sizeof(TIN) - 1)) == 0);
    const bool y_pairs_ok = (sd % 2 == 0) && (((uintptr_t)fa.y_out & (2 * sizeof(TIN) - 1)) == 0);
    const TIN *mean_b = (const TIN *)p.mean + (size_t)b * Tmax * ldi;
    const TIN *targ_b = (const TIN *)fa.target + (size_t)b * Tmax * sd;
    const int f0 = lane * M;
    unsigned long long liveS = 0ull, liveD = 0ull;
#pragma unroll
    for (int i = -1; i <= M; ++i) {
      const int t = f0 + i;
      if (t >= 0 && t < T) liveS |= 1ull << (i + 1);
      if (mw != 0 && t >= mw && t < T - mw) liveD |= 1ull << (i + 1);
    }
    const int baseR = RL::idx(f0, wv);
    const int baseD = DL::idx(f0, wv);
    const int loD = DL::idx(f0 > 0 ? f0 - 1 : 0, wv), hiD = DL::idx(f0 + M < 64 * M ? f0 + M : 64 * M - 1, wv);
    const int loR = RL::idx(f0 > 0 ? f0 - 1 : 0, wv), hiR = RL::idx(f0 + M < 64 * M ? f0 + M : 64 * M - 1, wv);

    double Pd[M], P1[M], P2[M], rhs[M];
    TIN *tileM = (TIN *)tileB;
    // one window's contribution to the matrix (unit precisions on the live frames) and, with means, to the right-hand side
    auto accumulate = [&](const int w, const bool with_mean, const TIN (&rm)[M + 2]) __attribute__((always_inline)) {
      const int l = ws.l[w], u = ws.u[w];
      const double *cw = ws.c + ws.off[w];
      const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
      const double c00 = c0 * c0, cpp = cp * cp, cmm = cm * cm, cp0 = cp * c0, c0m = c0 * cm, cpm = cp * cm;
      const unsigned long long live = w ? liveD : liveS;
#pragma unroll
      for (int i = -1; i <= M; ++i) {
        const bool lv = (live >> (i + 1)) & 1ull;
        const double tau = lv ? 1.0 : 0.0;
        double tm = 0.0;
        if (with_mean) tm = lv ? (double)rm[i + 1] : 0.0;
        if (i >= 0 && i < M) {
          Pd[i] += c00 * tau;
          P1[i] += cp0 * tau;
          if (with_mean) rhs[i] += c0 * tm;
        }
        if (i + 1 >= 0 && i + 1 < M) {
          Pd[i + 1] += cpp * tau;
          if (with_mean) rhs[i + 1] += cp * tm;
        }
        if (i - 1 >= 0 && i - 1 < M) {
          Pd[i - 1] += cmm * tau;
          P1[i - 1] += c0m * tau;
          P2[i - 1] += cpm * tau;
          if (with_mean) rhs[i - 1] += cm * tm;
        }
      }
    };
    auto fix_edges = [&]() __attribute__((always_inline)) {
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
    };
    // a (T, G) column group of a row-major array into the tile (asynchronous in DMA mode) ...
    auto issue = [&](TIN *tile, const TIN *src, const int ld) __attribute__((always_inline)) {
      if (DMA) load_tile_dma<M, TIN>(tile, src, ld, T, gvalid, wv, lane);
      else load_tile_regs<M, TIN>(tile, src, ld, T, gvalid, tid);
    };
    // ... and this lane's M (+2) values out of it, once it has landed; the tile is free again on return
    auto take = [&](const TIN *tile, TIN (&r)[M + 2]) __attribute__((always_inline)) {
      __syncthreads();
      if (DMA) {
        r[0] = tile[loD];
        r[M + 1] = tile[hiD];
#pragma unroll
        for (int i = 0; i < M; ++i) r[i + 1] = tile[baseD + i * DL::ESTRIDE];
      } else {
        r[0] = tile[loR];
        r[M + 1] = tile[hiR];
#pragma unroll
        for (int i = 0; i < M; ++i) r[i + 1] = tile[baseR + i];
      }
      __syncthreads();
    };
    constexpr bool kPark = (M >= 16) && MLPG_WAVE_PARK;  // the solver parks its multipliers in the tiles
    constexpr bool kPrefetchTarget = !kPark;              // otherwise the target tile travels under the forward solve

    // ---- forward: y = P^-1 sum_w W_w^T mu_w; the next window's tile is in flight under this window's arithmetic ----
#pragma unroll
    for (int i = 0; i < M; ++i) Pd[i] = P1[i] = P2[i] = rhs[i] = 0.0;
    issue(tileM, mean_b + d0, ldi);
    for (int w = 0; w < nw; ++w) {
      TIN rm[M + 2];
      take(tileM, rm);
      if (w + 1 < nw) issue(tileM, mean_b + (w + 1) * sd + d0, ldi);
      else if (kPrefetchTarget) issue(tileM, targ_b + d0, sd);
      accumulate(w, true, rm);
    }
    fix_edges();
    double *parkA = (double *)tileA + wv * RL::TPAD + lane * (M + kSkew), *parkB = (double *)tileB + wv * RL::TPAD + lane * (M + kSkew);
    // up to 8 frames per lane the cyclic reduction's blocks stay in registers and the backward solve is a second
    // right-hand side for the factorised matrix (no re-forming, no matrix arithmetic, a fifth of the shuffles)
    constexpr bool kKeep = (M <= 8) && !kPark && MLPG_FUSED_KEEP;
    PcrKeep keep;
    bool bad = solve_chunk<M, kPark, kKeep>(Pd, P1, P2, rhs, lane, parkA, parkB, &keep);
    int status = 0;
    if ((__ballot(bad) != 0ull) && sys_valid) {  // (cannot happen for window sets with a static window; kept for the contract)
      if (lane == 0) {
        const SysView<TIN, false> view = make_view<TIN, false>(p, ws, b, d, T);
        status = first_bad_pivot<2, TIN, false>(view, ws);
      }
      status = __shfl(status, 0);
    }
    if (sys_valid && lane == 0 && p.status) p.status[(size_t)b * p.ld_status + d] = status;
    const bool zero_out = status != 0;
    TIN rt[M + 2];
    if (!kPrefetchTarget) {
      __syncthreads();  // the solver is done with the tiles
      issue(tileM, targ_b + d0, sd);
    }
    take(tileM, rt);
    if (fa.y_out) {
      TIN *tileO = (TIN *)tileA;
#pragma unroll
      for (int i = 0; i < M; ++i)
        if ((liveS >> (i + 1)) & 1ull) tileO[baseR + i] = zero_out ? (TIN)0 : (TIN)rhs[i];
      __syncthreads();
      store_tile<M, TIN>(tileO, (TIN *)fa.y_out + (size_t)b * Tmax * sd + d0, sd, T, Tmax, gvalid, tid, y_pairs_ok);
      __syncthreads();
    }

    // ---- loss gradient in registers: g = scale * (y - target), loss += (y - target)^2 ----
    {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const bool lv = ((liveS >> (i + 1)) & 1ull) && sys_valid && !zero_out;
        const double y = (double)(TIN)rhs[i];  // the trajectory as the caller sees it (rounded to the output dtype)
        const double e = lv ? y - (double)rt[i + 1] : 0.0;
        lsum += e * e;
        rhs[i] = fa.scale * e;
      }
    }
    // ---- backward: z = P^-1 g with the same matrix ----
    if (kKeep) {
      solve_again<M>(Pd, P1, P2, rhs, lane, keep);
    } else {
      // re-formed from the windows alone and factorised again
#pragma unroll
      for (int i = 0; i < M; ++i) Pd[i] = P1[i] = P2[i] = 0.0;
      {
        const TIN none[M + 2] = {};
        for (int w = 0; w < nw; ++w) accumulate(w, false, none);
      }
      fix_edges();
      __syncthreads();  // the tiles double as the solver's parking area
      bad = solve_chunk<M, kPark>(Pd, P1, P2, rhs, lane, parkA, parkB);
      (void)bad;
    }
    // grad[t, w*sd+d] = tau_w[t] * (cm z[t-1] + c0 z[t] + cp z[t+1])   (paramgen/_mlpg.py:202-281, unit variances)
    double xl = __shfl_up(rhs[M - 1], 1), xr = __shfl_down(rhs[0], 1);
    if (lane == 0) xl = 0.0;
    if (lane == 63) xr = 0.0;
    TIN *tileO = (TIN *)tileB;
    for (int w = 0; w < nw; ++w) {
      const int l = ws.l[w], u = ws.u[w];
      const double *cw = ws.c + ws.off[w];
      const double cm = l ? cw[0] : 0.0, c0 = cw[l], cp = u ? cw[l + 1] : 0.0;
      const unsigned long long live = w ? liveD : liveS;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if ((liveS >> (i + 1)) & 1ull) {
          const double tau = ((live >> (i + 1)) & 1ull) ? 1.0 : 0.0;
          const double xm = (i == 0) ? xl : rhs[i > 0 ? i - 1 : 0];
          const double xp = (i == M - 1) ? xr : rhs[i < M - 1 ? i + 1 : M - 1];
          tileO[baseR + i] = zero_out ? (TIN)0 : (TIN)(tau * (cm * xm + c0 * rhs[i] + cp * xp));
        }
      }
      __syncthreads();
      store_tile<M, TIN>(tileO, (TIN *)p.out + (size_t)b * Tmax * ldo + w * sd + d0, ldo, T, Tmax, gvalid, tid, grad_pairs_ok);
    }
  }

  // ---- loss: lanes -> wavefront -> workgroup -> (last workgroup) the whole launch, every sum in a fixed order ----
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o);
  if (lane == 0) wsum[wv] = lsum;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int k = 0; k < kG; ++k) s += wsum[k];
    // agent-scope (write-through) store, drained, then the arrival -- no fence: an agent-scope fence on this chip writes
    // back and invalidates the whole L2 of the XCD, 960 times per launch (the first version: 130 us instead of 45)
    __hip_atomic_store((unsigned long long *)(fa.partials + blockIdx.x), (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(fa.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wsum[0] = (arrived == gridDim.x - 1) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (wsum[0] != 0.0 && wv == 0) {
    // the last workgroup: lane k sums partials k, k + 64, ... (fixed order), then a fixed shuffle tree
    double s = 0.0;
    for (unsigned q0 = 0; q0 < gridDim.x; q0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned q = q0 + k * 64 + lane;
        v[k] = q < gridDim.x ? __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)(fa.partials + q),
                                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                             : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      *fa.loss = s * fa.inv_n;
      __hip_atomic_store(fa.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
  }
}

template <int M, typename TIN, bool DMA>
int launch_fused_k(hipStream_t st, const Problem &p, const WinSet &ws, const FusedArgs &fa, int ngrp, int nslots) {
  constexpr size_t lds = 2 * (size_t)tile_bytes<M, TIN>();
  auto kern = wave_fused_kernel<M, TIN, DMA>;
  MLPG_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  note_launch(kCountFused);
  hipLaunchKernelGGL(kern, dim3(nslots * 8), dim3(kG * 64), lds, st, p, ws, ngrp, nslots, fa);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename TIN>
bool fused_dma_ok(const Problem &p, const FusedArgs &fa) {
  constexpr int epl = 16 / (int)sizeof(TIN);
  if (MLPG_WAVE_DMA == 0) return false;
  if (kG % epl || p.ld_in % epl || p.sd % epl) return false;
  return (((uintptr_t)p.mean | (uintptr_t)fa.target) & 15) == 0;
}

template <int M, typename TIN>
int launch_fused_m(hipStream_t st, const Problem &p, const WinSet &ws, const FusedArgs &fa, int ngrp, int nslots) {
  if (fused_dma_ok<TIN>(p, fa)) return launch_fused_k<M, TIN, true>(st, p, ws, fa, ngrp, nslots);
  return launch_fused_k<M, TIN, false>(st, p, ws, fa, ngrp, nslots);
}

template <typename TIN>
int launch_fused_t(hipStream_t st, const Problem &p, const WinSet &ws, const FusedArgs &fa, int ngrp, int nslots) {
  if (p.Tmax <= 64 * 4) return launch_fused_m<4, TIN>(st, p, ws, fa, ngrp, nslots);
  if (p.Tmax <= 64 * 8) return launch_fused_m<8, TIN>(st, p, ws, fa, ngrp, nslots);
  return launch_fused_m<16, TIN>(st, p, ws, fa, ngrp, nslots);
}

}  // namespace

bool unit_mse_supported(int Tmax, const WinSet &ws) {
  if (Tmax < 1 || Tmax > 64 * 16) return false;  // the 32-frames-per-lane instantiation has no registers to spare for two solves
  for (int w = 0; w < ws.nw; ++w)
    if (ws.l[w] > 1 || ws.u[w] > 1) return false;
  return true;
}

// Workspace (caller-owned, so that the call allocates nothing and can be captured into a graph): the arrival counter on
// the first 128-byte line (a fixed place: one workspace serves calls of different shapes), then one double per workgroup.  The counter must be zero when the kernel starts; the
// kernel leaves it zero, so the caller zeroes a fresh workspace ONCE.
size_t unit_mse_workspace_bytes(int B, int sd) {
  const size_t nblocks = (size_t)((B + 7) / 8) * ((sd + kG - 1) / kG) * 8;
  return (nblocks * sizeof(double) + 127) / 128 * 128 + 128;
}

int launch_unit_mse(hipStream_t st, int dtype, const Problem &p, const WinSet &ws, const void *target, void *y_out, double n_elems,
                    double *loss, void *workspace) {
  const int ngrp = (p.sd + kG - 1) / kG;
  const int nslots = ((p.B + 7) / 8) * ngrp;
  const size_t nblocks = (size_t)nslots * 8;
  char *sc = (char *)workspace;
  FusedArgs fa;
  fa.target = target;
  fa.y_out = y_out;
  fa.scale = 2.0 / n_elems;
  fa.inv_n = 1.0 / n_elems;
  (void)nblocks;
  fa.counter = (unsigned *)sc;
  fa.partials = (double *)(sc + 128);
  fa.loss = loss;
  return dtype == MLPG_HIP_F32 ? launch_fused_t<float>(st, p, ws, fa, ngrp, nslots) : launch_fused_t<double>(st, p, ws, fa, ngrp, nslots);
}

}  // namespace mlpg
