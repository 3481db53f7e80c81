// Walk form of the strip MLPG kernel (round 6): ONE workgroup walks the strips of an utterance in order.
//
// Same mapping and the same three-level substructured LDL^T as mlpg_strip_impl.h -- lane = static dim, wavefront = chunk of
// 16 frames, workgroup = strip of 4 chunks; levels 1 and 2 are that header's code -- but the strips of an utterance are not
// dealt to different workgroups: one workgroup (one per CU: the previous strip's chunk factors stay parked in registers
// while the next strip's level 1 runs, 512 registers per lane at one wavefront per SIMD) takes them one after the other.
// Level 3 then needs no exchange between workgroups at all (tools/walk_model.py is the executable specification, pinned
// against the oracle by tests/test_walk_model.py):
//   * the top-down elimination over the strips' last separators is carried EXACTLY from strip to strip (row j is
//     finalised when record j + 1 is at hand: A_j = E_j - T_{j+1} - M_j V_j^T, a_j = g_j - h_{j+1} - M_j a_{j-1},
//     M_{j+1} = V_{j+1} A_j^-1);
//   * strip r is finished one step late, when record r + 1 exists: sigma_r = A_r^-1 a_r with separator r + 1 clamped to
//     zero (the strip kernel's 3-strip window, exact on its upper side), sigma_{r-1} = A_{r-1}^-1 (a_{r-1} - V_r^T sigma_r);
//     accepted if 2 max|A_r^-1 V_{r+1}^T| max(1, 2 max|A_{r-1}^-1 V_r^T|) < 2^-66 for every system, else the utterance is
//     MARKED and left to the strip kernel's general route (launched behind this kernel, it only takes marked utterances);
//   * the last strip is exact.
// Why (profiles/r06_notes.md section 3): in the strip kernel the workgroups that hold an utterance's strips wait for each
// other's records (6.8 us of an item's 25.5 us; 3.9 of 15.9 us even at one workgroup per CU) and draw a ticket per item
// (2 us); here an item is level 1 (7.2 us at one wavefront per SIMD) plus 2-3 us of levels 2-3 out of LDS, back-substitution
// and stores.  Utterance-granular: worth it when the launch has about a multiple of the CU count of utterances.
#pragma once
#include "mlpg_strip_impl.h"

#ifndef MLPG_WALK_RING
#define MLPG_WALK_RING 6  // frames of loads in flight per wavefront in the walk form's level 1 (interior chunks)
#endif

namespace mlpg {
namespace walk {
using namespace strip;

constexpr size_t kWLdsRec = (size_t)kW * kRec * 64 * 8;
constexpr size_t kWLdsFac = (size_t)2 * (kW - 1) * kFac * 64 * 8;  // two strips' level-2 factors (the held strip's and the new one's)
constexpr size_t kWLdsU = (size_t)((kW + 1) * 2 + 1) * 64 * 8;
constexpr size_t kWLdsMisc = 256;
constexpr size_t kWLdsBytes = kWLdsRec + kWLdsFac + kWLdsU + kWLdsMisc;
inline size_t walk_lds_bytes() { return kWLdsBytes; }

template <typename TIN, typename TOUT, int VM>
__global__ __launch_bounds__(kW * 64, 1) void walk_kernel(Problem p, WinSet ws, strip::Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *lds_rec = (double *)smem;                                  // [kW][kRec][64]
  double *lds_fac = (double *)(smem + kWLdsRec);                     // [2][kW-1][kFac][64]
  double *lds_u = (double *)(smem + kWLdsRec + kWLdsFac);            // [kW+1][2][64] + [64]: slot 0 = previous strip's last separator
  int *lds_misc = (int *)(smem + kWLdsRec + kWLdsFac + kWLdsU);      // [0] group ticket, [1] reject
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Tmax = p.Tmax;
  const long ldi = p.ld_in, ldo = p.ld_out;
  const int sd = p.sd, mw = ws.mw;
  const int nd = sd < 64 ? sd : 64;
  const bool lane_ok = lane < nd;
  const int d = lane_ok ? lane : nd - 1;  // idle lanes shadow the last dim (never stored)
  const unsigned loff = (unsigned)d * (unsigned)sizeof(TIN);
  int *const ticket = a.ctrl + kCtrlWalkTicket;
  if (a.stagger > 0) {
    // start ramp: the workgroups do not depend on each other, so a spread start stays spread -- at any time the same share of them
    // is in level 1 (loading) instead of all of them at once
    const long long until = (long long)__builtin_amdgcn_s_memrealtime() + (long long)blockIdx.x * a.stagger / (long long)gridDim.x;
    while ((long long)__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  }

  for (;;) {
    if (tid == 0) lds_misc[0] = atomicAdd(ticket, 1);
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(lds_misc[0]);
    __syncthreads();  // (lds_misc[0] is rewritten at the top of the next round)
    if (g >= a.nsg) break;
    const int b = g;  // one dim group per utterance
    int T = p.lengths ? p.lengths[b] : Tmax;
    T = T < 0 ? 0 : (T > Tmax ? Tmax : T);
    const int Ract = (T + kW * kM - 1) / (kW * kM);
    TOUT *out_b = (TOUT *)p.out + (size_t)b * Tmax * ldo;
    const __amdgpu_buffer_rsrc_t ors = make_rsrc(out_b);
    const unsigned ooff = (unsigned)d * (unsigned)sizeof(TOUT), ldo_b = (unsigned)ldo * (unsigned)sizeof(TOUT);
    int *line = a.ctrl + (1 + kMaxLists + g) * kCtrlLine;
    if (wv == 0 && lane_ok && p.status) p.status[(size_t)b * p.ld_status + d] = 0;  // (verdict_kernel overrides it for marked systems)
    const __amdgpu_buffer_rsrc_t mrs = make_rsrc((const TIN *)p.mean + (size_t)b * Tmax * ldi);
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(VM == MLPG_HIP_VAR_FRAME ? (const TIN *)p.var + (size_t)b * Tmax * ldi : (const TIN *)p.out);
    const TIN *vglob = VM == MLPG_HIP_VAR_GLOBAL ? (const TIN *)p.var + d : nullptr;
    const __amdgpu_buffer_rsrc_t grs = make_rsrc(p.out);

    // the held strip's chunk factor (this wavefront's chunk) and, in wavefront 0, the carried state of level 3
    double PdP[kN], P1P[kN], P2P[kN], gP[kN], caP = 0.0, cbP = 0.0, ccP = 0.0;
    S2 Ainv_p = {1.0, 0.0, 1.0};   // A_{r-1}^-1 of the last finalised row
    V2 av_p = {0.0, 0.0};          // a_{r-1}
    M2 Mn = {0.0, 0.0, 0.0, 0.0};  // M_r = V_r A_{r-1}^-1 for the held strip r
    S2 E_h = {1.0, 0.0, 1.0};      // the held strip's record: E, g, V
    V2 g_h = {0.0, 0.0};
    M2 V_h = {0.0, 0.0, 0.0, 0.0};
    unsigned long long bad_mask = 0ull;
    bool rejected = false;

    // back-substitution of the held strip's chunk and its 16 trajectory rows
    auto finish_chunk = [&](const int rh) __attribute__((always_inline)) {
      const V2 ul = {lds_u[(wv * 2) * 64 + lane], lds_u[(wv * 2 + 1) * 64 + lane]};
      const V2 uo = {lds_u[((wv + 1) * 2) * 64 + lane], lds_u[((wv + 1) * 2 + 1) * 64 + lane]};
      const double sx = lds_u[((kW + 1) * 2) * 64 + lane];
      const bool sys_bad = !(sx == sx) || !(uo.x == uo.x) || !(ul.x == ul.x);  // NaN: some pivot of this system failed
      double Pd[kM], P1[kM], P2[kM], rhs[kM];
#pragma unroll
      for (int i = 0; i < kN; ++i) { Pd[i] = PdP[i]; P1[i] = P1P[i]; P2[i] = P2P[i]; rhs[i] = gP[i]; }
      Pd[kN] = Pd[kN + 1] = 1.0; P1[kN] = P1[kN + 1] = P2[kN] = P2[kN + 1] = 0.0; rhs[kN] = rhs[kN + 1] = 0.0;
      backsub(Pd, P1, P2, rhs, caP, cbP, ccP, ul, uo);
      if (wv == 0) {
        const unsigned long long m = __ballot(sys_bad && lane_ok);
        bad_mask |= m;
      }
      if (!lane_ok) return;
      const int f0 = (rh * kW + wv) * kM;
#pragma unroll
      for (int i = 0; i < kM; ++i) {
        const int t = f0 + i;
        if (t < Tmax) st_row(ors, (unsigned)t * ldo_b, ooff, (t < T && !sys_bad) ? (TOUT)rhs[i] : (TOUT)0);
      }
    };

    for (int r = 0; r <= Ract && !rejected; ++r) {
      double Pd[kM], P1[kM], P2[kM], rhs[kM], ca = 0.0, cb = 0.0, cc = 0.0;
      if (r < Ract) {
        // ---- level 1 of strip r ----
        double rec[kRec];
        bool bad = false;
        const int f0 = (r * kW + wv) * kM;
        if (f0 < T) {
          const bool interior = mw != 0 && f0 - 1 >= mw && f0 + kM < T - mw;
          double wcl[3][9];
          karg_f64x9<kKargWc + 0 * 72>(wcl[0]);
          karg_f64x9<kKargWc + 1 * 72>(wcl[1]);
          karg_f64x9<kKargWc + 2 * 72>(wcl[2]);
          float tk[kM + 1][3];
          if (interior) bad = assemble_eliminate<TIN, false, VM, false, 3, false, false, false, MLPG_WALK_RING>(mrs, vrs, grs, vglob, loff, ldi, 0, sd, f0, T, mw, wcl, a.one, Pd, P1, P2, rhs, ca, cb, cc, rec, tk);
          else bad = assemble_eliminate<TIN, false, VM, true, 3>(mrs, vrs, grs, vglob, loff, ldi, 0, sd, f0, T, mw, wcl, a.one, Pd, P1, P2, rhs, ca, cb, cc, rec, tk, T);
        } else {
          // a chunk of identity rows behind the utterance's end (keeps the strip's separator chain regular)
#pragma unroll
          for (int i = 0; i < kM; ++i) { Pd[i] = 1.0; P1[i] = P2[i] = rhs[i] = 0.0; }
#pragma unroll
          for (int k = 0; k < kRec; ++k) rec[k] = 0.0;
          rec[rD11] = rec[rD22] = 1.0;
        }
        if (bad) rec[rD11] = __builtin_nan("");  // poisons every later level: the system is reported, not solved
#pragma unroll
        for (int k = 0; k < kRec; ++k) lds_rec[(wv * kRec + k) * 64 + lane] = rec[k];
      }
      __syncthreads();
      if (wv == 0) {
        S2 E = {1.0, 0.0, 1.0}, Ts = {0.0, 0.0, 0.0};
        V2 gg = {0.0, 0.0}, hs = {0.0, 0.0};
        M2 V = {0.0, 0.0, 0.0, 0.0};
        if (r < Ract) {
          // ---- level 2 of strip r (as the strip kernel's): the strip's record E, g, V, T, h; the separators' factors to LDS ----
          auto R_ = [&](int j, int k) __attribute__((always_inline)) { return lds_rec[(j * kRec + k) * 64 + lane]; };
          E = {R_(0, rD11), R_(0, rD12), R_(0, rD22)};
          gg = {R_(0, rF1), R_(0, rF2)};
          V = {R_(0, rL11), R_(0, rL12), R_(0, rL21), R_(0, rL22)};
          if (r == 0) V = {0.0, 0.0, 0.0, 0.0};
          Ts = {R_(0, rT00), R_(0, rT01), R_(0, rT11)};
          hs = {R_(0, rH0), R_(0, rH1)};
          E = sub(E, S2{R_(1, rT00), R_(1, rT01), R_(1, rT11)});
          gg = sub(gg, V2{R_(1, rH0), R_(1, rH1)});
          bool bad2 = false;
#pragma unroll
          for (int j = 0; j + 1 < kW; ++j) {
            const S2 Einv = sym_inv(E, bad2);
            const M2 L = {R_(j + 1, rL11), R_(j + 1, rL12), R_(j + 1, rL21), R_(j + 1, rL22)};
            const M2 Mj = mul_ms(L, Einv);
            const M2 EV = mul_sm(Einv, V);
            const V2 c = mul_sv(Einv, gg);
            Ts = add(Ts, mul_mtm_sym(V, EV));
            hs = add(hs, mul_mtv(V, c));
            double *f = lds_fac + ((size_t)(r & 1) * (kW - 1) + j) * kFac * 64 + lane;
            f[0 * 64] = c.x; f[1 * 64] = c.y;
            f[2 * 64] = EV.a; f[3 * 64] = EV.b; f[4 * 64] = EV.c; f[5 * 64] = EV.d;
            f[6 * 64] = Mj.a; f[7 * 64] = Mj.b; f[8 * 64] = Mj.c; f[9 * 64] = Mj.d;
            S2 Dn = {R_(j + 1, rD11), R_(j + 1, rD12), R_(j + 1, rD22)};
            V2 Fn = {R_(j + 1, rF1), R_(j + 1, rF2)};
            if (j + 2 < kW) {
              Dn = sub(Dn, S2{R_(j + 2, rT00), R_(j + 2, rT01), R_(j + 2, rT11)});
              Fn = sub(Fn, V2{R_(j + 2, rH0), R_(j + 2, rH1)});
            }
            E = sub(Dn, mul_mmt_sym(Mj, L));
            gg = sub(Fn, mul_mv(Mj, gg));
            V = neg(mul_mm(Mj, V));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (bad2) E.a = __builtin_nan("");
        }
        if (r > 0) {
          // ---- level 3 for the held strip rh = r - 1: finalise its row with this strip's T, h (none behind the last strip) ----
          const int rh = r - 1;
          S2 A = E_h;
          V2 aa = g_h;
          if (r < Ract) {
            A = sub(A, Ts);
            aa = sub(aa, hs);
          }
          if (rh > 0) {
            A = sub(A, mul_mmt_sym(Mn, V_h));
            aa = sub(aa, mul_mv(Mn, av_p));
          }
          bool bad3 = false;
          const S2 Ainv = sym_inv(A, bad3);
          V2 sig = mul_sv(Ainv, aa);
          if (bad3) sig.x = __builtin_nan("");
          V2 sprev = {0.0, 0.0};
          double damp = 0.0;
          if (r < Ract) damp = 2.0 * amax4(mul_smt(Ainv, V));  // what clamping separator r ignores reaches row rh through A^-1 V_r^T
          if (rh > 0) {
            sprev = sub(mul_sv(Ainv_p, av_p), mul_mtv(Mn, sig));
            damp *= __builtin_fmax(1.0, 2.0 * amax4(mul_smt(Ainv_p, V_h)));
          }
          // (a system with a failing pivot carries NaN: it is reported by verdict_kernel, not a reason for the general route)
          const bool over = lane_ok && damp == damp && !(damp < kDamp1Tol);
          const int rej = __ballot(over) != 0ull;
          if (lane == 0) lds_misc[1] = rej;
          // level-2 back-substitution of strip rh: s = sigma_{rh-1}, u_last = sigma_rh
          const double *fb = lds_fac + (size_t)(rh & 1) * (kW - 1) * kFac * 64 + lane;
          V2 un = sig;
          lds_u[0 * 64 + lane] = sprev.x;
          lds_u[1 * 64 + lane] = sprev.y;
          lds_u[(kW * 2) * 64 + lane] = sig.x;
          lds_u[(kW * 2 + 1) * 64 + lane] = sig.y;
#pragma unroll
          for (int j = kW - 2; j >= 0; --j) {
            const double *f = fb + (size_t)j * kFac * 64;
            const V2 c = {f[0 * 64], f[1 * 64]};
            const M2 EV = {f[2 * 64], f[3 * 64], f[4 * 64], f[5 * 64]};
            const M2 Mj = {f[6 * 64], f[7 * 64], f[8 * 64], f[9 * 64]};
            const V2 u = sub(sub(c, mul_mv(EV, sprev)), mul_mtv(Mj, un));
            lds_u[((j + 1) * 2) * 64 + lane] = u.x;
            lds_u[((j + 1) * 2 + 1) * 64 + lane] = u.y;
            un = u;
          }
          lds_u[((kW + 1) * 2) * 64 + lane] = sig.x + sprev.x;  // NaN if the system failed anywhere so far
          // carried state for the next row
          Mn = mul_ms(V, Ainv);
          Ainv_p = Ainv;
          av_p = aa;
        } else if (lane == 0) {
          lds_misc[1] = 0;
        }
        E_h = E;
        g_h = gg;
        V_h = V;
      }
      __syncthreads();
      if (r > 0) {
        rejected = __builtin_amdgcn_readfirstlane(lds_misc[1]) != 0;
        if (!rejected) finish_chunk(r - 1);
      }
      if (r < Ract) {
        // park this strip's chunk factor
#pragma unroll
        for (int i = 0; i < kN; ++i) { PdP[i] = Pd[i]; P1P[i] = P1[i]; P2P[i] = P2[i]; gP[i] = rhs[i]; }
        caP = ca; cbP = cb; ccP = cc;
      }
    }
    // the utterance's padding frames behind its last strip
    if (!rejected && lane_ok) {
      for (int t = Ract * kW * kM + wv * kM; t < Tmax; t += kW * kM)
        for (int i = 0; i < kM && t + i < Tmax; ++i) st_row(ors, (unsigned)(t + i) * ldo_b, ooff, (TOUT)0);
    }
    if (wv == 0 && lane == 0) {
      if (rejected) {
        __hip_atomic_store(line + kLineMarked, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.ctrl + kCtrlMarked, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (bad_mask != 0ull) {
        __hip_atomic_fetch_or(line + 2, (int)(unsigned)bad_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(line + 3, (int)(unsigned)(bad_mask >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace walk
}  // namespace mlpg
