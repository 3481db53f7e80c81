// Launcher of the chunked MLPG kernel (window extents up to 2): see mlpg_chunk_impl.h.
#include <cstdlib>

#include "mlpg_chunk_impl.h"

namespace mlpg {

using namespace chunk;

bool chunk_supported(const Problem &p, const WinSet &ws) {
  if (ws.nw < 1 || ws.nw > kMaxNw || ws.mw < 1 || ws.mw > 2) return false;
  if (p.pitch && p.pitch != p.sd) return false;
  return rows_fit_buffer(p);
}

// the natural-order kernel is the only other one that takes extents of 2; a handful of systems are not worth three launches
bool chunk_preferred(const Problem &p, const WinSet &ws, bool backward) {
  (void)backward;
  return chunk_supported(p, ws) && ws.mw == 2 && (long)p.B * p.sd >= 64 && p.Tmax >= 64;
}

namespace {

template <int Q>
void fill_args(const Problem &p, const WinSet &ws, Args *a) {
  constexpr int EXT = Geo<Q>::EXT;
  a->nw = ws.nw;
  a->mw = ws.mw;
  a->wspec = (ws.nw == 3 && ws.l[0] == 0 && ws.u[0] == 0 && (ws.l[1] | ws.u[1]) && (ws.l[2] | ws.u[2])) ? 1 : 0;
  for (int w = 0; w < kMaxNw; ++w) {
    a->narrow[w] = 0;
    for (int j = 0; j < 5; ++j) a->cpad[w][j] = 0.0;
    if (w >= ws.nw) continue;
    const int l = ws.l[w], u = ws.u[w];
    a->narrow[w] = (l == 0 && u == 0) ? 1 : 0;
    // W_w[t, t + j] = c_w[l + j], j in [-l, u]
    for (int j = -l; j <= u; ++j) a->cpad[w][j + EXT] = ws.c[ws.off[w] + l + j];
  }
}

// Measured and dropped (tools/gpurun/r4_chunk.sh history, profiles/r04_notes.md): slabs of utterances sized to stay in the Infinity
// Cache between pass 1 and pass 3 (48 / 96 / 192 MB: 2.10 / 1.15 / 0.84 ms against 0.58 for the whole batch: pass 2's ~120 us of
// sequential latency is paid per slab); pass 2 of one part of the batch on a side stream under pass 1 / pass 3 of the others
// (2 / 3 / 4 parts: 0.70 / 0.75 / 0.76 ms against 0.54: the chunk kernels book the whole register file, the side stream's
// wavefronts wait for them to drain).
template <typename TIN, typename TOUT, int Q, bool BWD>
int launch_q(hipStream_t st, const Problem &p, const WinSet &ws, int device) {
  using G = Geo<Q>;
  Args a;
  fill_args<Q>(p, ws, &a);
  a.dgw = p.sd < 64 ? p.sd : 64;
  a.ndg = (p.sd + a.dgw - 1) / a.dgw;
  a.K = (p.Tmax + G::C - 1) / G::C;
  a.nsg = p.B * a.ndg;
  const size_t nsg = (size_t)a.nsg;
  const size_t rec_b = nsg * a.K * G::kRec * 64 * sizeof(double), fac_b = nsg * a.K * G::kFac * 64 * sizeof(double),
               xs_b = nsg * a.K * Q * 64 * sizeof(double), mid_b = nsg * G::kFac * 64 * sizeof(double), bad_b = nsg * 64 * sizeof(int);
  char *sc = (char *)scratch(device, st, 6, rec_b + fac_b + xs_b + mid_b + bad_b + 256);
  if (!sc) return MLPG_HIP_ENOMEM;
  a.rec = (double *)sc;
  a.fac = (double *)(sc + rec_b);
  a.xs = (double *)(sc + rec_b + fac_b);
  a.mid = (double *)(sc + rec_b + fac_b + xs_b);
  a.bad = (int *)(sc + rec_b + fac_b + xs_b + mid_b);
  MLPG_HIP_CHECK(hipMemsetAsync(a.bad, 0, bad_b, st));  // the marks of non-positive pivots
  constexpr size_t lds3 = (size_t)kW * (G::NLDS ? G::NLDS : 0) * (Q + 1) * 64 * sizeof(double);
  constexpr size_t lds1 = 0;
  const long items = (long)a.nsg * a.K;
  const dim3 grid((unsigned)((items + kW - 1) / kW)), block(kW * 64);
  note_launch(kCountChunk);
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_FRAME, Q, false, BWD>), grid, block, lds1, st, p, a); break;
    case MLPG_HIP_VAR_GLOBAL: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_GLOBAL, Q, false, BWD>), grid, block, lds1, st, p, a); break;
    default: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_UNIT, Q, false, BWD>), grid, block, lds1, st, p, a);
  }
  hipLaunchKernelGGL((reduce_elim_kernel<Q>), dim3((unsigned)(2 * a.nsg)), dim3(64), 0, st, p, a);
  hipLaunchKernelGGL((reduce_subst_kernel<Q>), dim3((unsigned)(2 * a.nsg)), dim3(64), 0, st, p, a);
  switch (p.var_mode) {
    case MLPG_HIP_VAR_FRAME: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_FRAME, Q, true, BWD>), grid, block, lds3, st, p, a); break;
    case MLPG_HIP_VAR_GLOBAL: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_GLOBAL, Q, true, BWD>), grid, block, lds3, st, p, a); break;
    default: hipLaunchKernelGGL((chunk_kernel<TIN, TOUT, MLPG_HIP_VAR_UNIT, Q, true, BWD>), grid, block, lds3, st, p, a);
  }
  hipLaunchKernelGGL((verdict_kernel<TIN, TOUT, Q, BWD>), dim3((unsigned)(((long)p.B * p.sd + 255) / 256)), dim3(256), 0, st, p, ws, a);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename TIN, typename TOUT>
int launch_t(hipStream_t st, bool backward, const Problem &p, const WinSet &ws, int device) {
  if (backward) return ws.mw == 2 ? launch_q<TIN, TOUT, 4, true>(st, p, ws, device) : launch_q<TIN, TOUT, 2, true>(st, p, ws, device);
  return ws.mw == 2 ? launch_q<TIN, TOUT, 4, false>(st, p, ws, device) : launch_q<TIN, TOUT, 2, false>(st, p, ws, device);
}

}  // namespace

int launch_chunk(hipStream_t st, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &ws, int device) {
  if (dtype != out_dtype || !chunk_supported(p, ws)) {
    set_error("MLPG_HIP_ALGO_CHUNK: input dtype = output dtype, 1-3 windows of extent 1 or 2");
    return MLPG_HIP_EINVAL;
  }
  return dtype == MLPG_HIP_F32 ? launch_t<float, float>(st, backward, p, ws, device) : launch_t<double, double>(st, backward, p, ws, device);
}

}  // namespace mlpg
