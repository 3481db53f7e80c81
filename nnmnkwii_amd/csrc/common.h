// Shared host/device definitions for libmlpg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mlpg_hip.h"

namespace mlpg {

constexpr int kMaxWindows = 8;
constexpr int kMaxCoef = 48;   // sum over windows of (l+u+1)
constexpr int kMaxExtent = 4;  // max l or u of any window (half-bandwidth <= 8)

// Window table passed BY VALUE as a kernel argument (lands in SGPRs / the
// kernarg segment; no device allocation, no __constant__ upload to order).
struct WinSet {
  int nw;
  int q;   // half-bandwidth of P: max_w (l_w + u_w)            (_mlpg.py:72-73)
  int mw;  // edge width: max_w max(l_w, u_w)                   (_mlpg.py:177)
  int l[kMaxWindows];
  int u[kMaxWindows];
  int off[kMaxWindows];  // offset of window w's coefficients in c[]
  double c[kMaxCoef];
};

// Problem description shared by the forward and backward kernels.
struct Problem {
  const void *mean;      // (B, Tmax, D)   forward only
  const void *var;       // per var_mode
  const void *grad_out;  // (B, Tmax, sd)  backward only
  const int32_t *lengths;
  void *out;             // forward: (B, Tmax, sd); backward: (B, Tmax, D)
  int32_t *status;
  int var_mode;
  int B, Tmax, D, sd;
  // Columns between a dim's windows in a row; 0 = sd.  Differs from sd when the problem is a PIECE of a stream (some of
  // its static dims: mlpg_hip_forward_streams cuts streams to fill lane groups): forward pass on the wave-per-system
  // and the natural-order kernels only.
  int pitch = 0;
  // Row strides in elements; the utterance stride is Tmax * row stride (the parent array is a
  // densely packed (B, Tmax, ld) batch of which this problem is a column slice: one stream of a
  // multi-stream acoustic feature matrix).  Dense problems: ld_in = D, ld_gout = sd,
  // ld_out = sd (forward) | D (backward), ld_status = sd.
  long ld_in;     // mean and per-frame var rows
  long ld_gout;   // grad_out rows (backward)
  long ld_out;    // out rows
  int ld_status;  // status[b * ld_status + d]
};

// Several streams of one (B, Tmax, ld) batch solved by ONE strip-kernel launch: their static dims sit side by side on
// the lanes (merged index begin[s] .. begin[s+1]).  Columns are absolute in the parent arrays; unused entries of
// begin[] are INT_MAX.
struct StreamMap {
  int n, total;
  int begin[4], in_col[4], sd[4], out_col[4], stat_col[4];
  // Transposed form (tr_u > 0; strip kernel, round 5): ONE narrow stream whose lanes run over tr_u consecutive UTTERANCES x its
  // tr_nd static dims (lane = u * tr_nd + d; total = tr_u * tr_nd <= 64), so that a stream of 1 .. 32 dims fills the 64 lanes
  // it would otherwise leave idle.  A system group is then a block of tr_u utterances (tr_B utterances in all: the last block may
  // be short); sd[0] is the window pitch, in_col[0] / out_col[0] / stat_col[0] the stream's columns, and tr_in / tr_out / tr_stat
  // the element strides from one utterance to the next in the input, output and status arrays.  With a lengths vector the group
  // runs to its longest utterance and every lane masks its own dead frames (strip::assemble_eliminate<..., LT>).
  int tr_u, tr_nd, tr_B, tr_in, tr_out, tr_stat;
};

// The strip, constant-coefficient and chunked kernels address an utterance's rows through a buffer descriptor with 32-bit byte
// offsets from the utterance's first row (2 GB window): longer utterances (Tmax * row stride * 8 bytes) go to the natural-order kernel.
inline bool rows_fit_buffer(const Problem &p) {
  const long ld = p.ld_in > p.ld_out ? (p.ld_in > p.ld_gout ? p.ld_in : p.ld_gout) : (p.ld_out > p.ld_gout ? p.ld_out : p.ld_gout);
  return (double)p.Tmax * (double)ld * 8.0 < 2147483647.0;
}

void set_error(const char *fmt, ...);
// launches per kernel family since the library was loaded (mlpg_hip_launch_count: a test aid)
enum { kCountGeneric = 0, kCountWave, kCountStrip, kCountStripMulti, kCountConst, kCountFused, kCountChunk, kCountFir, kCountConstMulti, kCountStripTr, kCountHostSmall, kCountHostSmallDirect, kCountKinds };
void note_launch(int kind);
// Grow-only scratch, cached per (device, stream, slot): slot 0 generic factor, 1 fastdtw pyramids,
// 2 generic status, 3 strip records, 4 constant-coefficient kernel (factor table), 5 fastdtw from host costs (D rows, back-pointers), 6 chunked kernel (records, block factors, separator solutions, marks).  Returns nullptr (and sets the error) on failure.
void *scratch(int device, hipStream_t stream, int slot, size_t bytes, unsigned long long *gen = nullptr);

// launchers (one per translation unit)
int launch_generic(hipStream_t s, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w,
                   int device);
bool wave_supported(const Problem &p, const WinSet &w);
int launch_wave(hipStream_t s, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w,
                int device);
bool strip_supported(const Problem &p, const WinSet &w);
bool strip_preferred(const Problem &p, const WinSet &w, bool backward, int in_dtype);
bool strip_tr_supported(const Problem &p, const WinSet &w, bool backward, int in_dtype, int out_dtype);
bool strip_tr_preferred(const Problem &p, const WinSet &w, bool backward, int in_dtype, int out_dtype);
int launch_strip_tr(hipStream_t s, int dtype, const Problem &p, const WinSet &w, int device);
int launch_strip(hipStream_t s, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w,
                 int device, bool try_tr = true);
// launch_strip_multi's "nothing was enqueued" result: fewer workgroups can be resident than an utterance has strips
constexpr int kStripMultiNotResident = -1000;
// forward pass of several streams (p: the parent arrays, sd/D unused) -- per-frame variances, three windows of extent <= 1
int launch_strip_multi(hipStream_t s, int dtype, const Problem &p, const WinSet &w, const StreamMap &sm, int device);
bool unit_mse_supported(int Tmax, const WinSet &w);
size_t unit_mse_workspace_bytes(int B, int sd);
int launch_unit_mse(hipStream_t s, int dtype, const Problem &p, const WinSet &w, const void *target, void *y_out,
                    double n_elems, double *loss, void *workspace);
bool const_supported(const Problem &p, const WinSet &w);
// true once per scratch allocation (device, stream): the constant-coefficient kernel's table holds nothing yet
bool const_scratch_fresh(int device, hipStream_t stream, unsigned long long gen);
// unit variances: true if this (device, stream) last built its table from the same key on the same scratch allocation
bool const_unit_table_cached(int device, hipStream_t stream, unsigned long long gen, bool fresh, const double *key, int n);
// strip kernel: do the workgroups of a launch on the current device spread evenly over eight XCDs (HW_REG_XCC_ID)?  Probed once
// per device with a small kernel (mlpg_strip.hip); false while `st` is being captured and the device has not been probed yet.
bool strip_xcd_lists_ok(hipStream_t st);
bool const_preferred(const Problem &p, const WinSet &w);
int launch_const(hipStream_t s, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w, int device);
int launch_const_multi(hipStream_t s, int dtype, const Problem &p, const WinSet &w, const StreamMap &sm, int device);
bool chunk_supported(const Problem &p, const WinSet &w);
bool chunk_preferred(const Problem &p, const WinSet &w, bool backward);
int launch_chunk(hipStream_t s, int dtype, int out_dtype, bool backward, const Problem &p, const WinSet &w, int device);
int launch_chunk_fwd_f64(hipStream_t s, const Problem &p, const WinSet &w, int device);
int launch_chunk_fwd_f32(hipStream_t s, const Problem &p, const WinSet &w, int device);
int launch_chunk_bwd_f64(hipStream_t s, const Problem &p, const WinSet &w, int device);
int launch_chunk_bwd_f32(hipStream_t s, const Problem &p, const WinSet &w, int device);
// unit variances on float32 tensors as a FIR filter; launch_fir returns kFirNotApplicable when the window set's inverse does not decay
// fast enough (or its tap table cannot be built now: stream capture)
constexpr int kFirNotApplicable = -2000;
bool fir_shape_supported(const Problem &p, const WinSet &w, int in_dtype, int out_dtype);
bool fir_preferred(const Problem &p, bool backward);
bool fir_table_ready(hipStream_t s, int device, const WinSet &w);
void fir_shutdown();
int launch_fir(hipStream_t s, bool backward, const Problem &p, const WinSet &w, int device);
size_t fir_mse_workspace_bytes(int B, int Tmax, int sd);
int launch_fir_mse(hipStream_t s, const Problem &p, const WinSet &w, int device, const void *target, void *y_out, double n_elems,
                   double *loss, void *workspace);
int launch_copy_cols(hipStream_t s, int dtype, const void *src, long ld_src, const int32_t *lengths, int B, int Tmax,
                     int ncols, void *dst, long ld_dst);
int launch_stream_copy(hipStream_t s, const void *src, void *dst, size_t nbytes);
int launch_modspec(hipStream_t s, int mode, const double *x, const double *ms, const double *ph, double *out,
                   double *out_ph, int B, int T, int D, int n, int ortho, int limit_bin, int log_domain);
void host_api_shutdown();  // host_api.hip: streams, events, pinned and device staging buffers of the _host entry points
int launch_modspec_dft(hipStream_t s, int device, int mode, const double *x, const double *ms, const double *ph,
                       double *out, double *out_ph, int B, int T, int D, int n, int ortho, int limit_bin,
                       int log_domain);
int launch_delta(hipStream_t s, int dtype, const void *x, const int32_t *lengths, int B, int Tmax, int D,
                 const WinSet &w, void *out);
int launch_trim(hipStream_t s, int dtype, const void *X, int N, int T, int D, double eps, int32_t *lengths);
int launch_fastdtw(hipStream_t s, int device, const double *X, const double *Y, const int32_t *lenx,
                   const int32_t *leny, int N, int Tx, int Ty, int D, int radius, int dist_kind, double dist_scale,
                   int tie_rule, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *cost);
int launch_dtw_window(hipStream_t s, int device, int N, int radius, const int32_t *ltx, const int32_t *lty,
                      const int32_t *full, const int32_t *cpath_i, const int32_t *cpath_j, const int32_t *cpath_len,
                      int cpath_stride, int32_t *row_lo, int32_t *row_hi, int64_t *row_off, int row_stride);
int launch_dtw_costs(hipStream_t s, int device, int N, int tie_rule, const int32_t *ltx, const int32_t *lty,
                     const int32_t *row_lo, const int32_t *row_hi, const int64_t *row_off, int row_stride, int max_ty,
                     const double *costs, const int64_t *cost_base, int64_t total_cells, int32_t *path_i, int32_t *path_j,
                     int32_t *path_len, int path_stride, double *cost_out);
int launch_gmm_convert(hipStream_t s, const double *x, const double *post, const int32_t *mix, const double *mu_x,
                       const double *mu_y, const double *A, long N, int D, int Dy, int M, double *out);
int launch_gather(hipStream_t s, int dtype, const void *src, const int32_t *path, const int32_t *path_len, int N,
                  int Tsrc, int path_stride, int D, int Tout, void *out);

#define MLPG_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      ::mlpg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MLPG_HIP_ERUNTIME;                                                     \
    }                                                                               \
  } while (0)

}  // namespace mlpg
