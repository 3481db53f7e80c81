/*
 * include/mlpg_hip.h -- C ABI of libmlpg_hip.so (MI355X / gfx950).
 *
 * The reference (r9y9/nnmnkwii) has no FFI for this path: its native code is a
 * set of Cython extension modules imported by name.  This header is therefore
 * the boundary *we* introduce beneath the reference's Python signatures; each
 * entry point names the reference routine(s) it replaces (paths relative to
 * /root/reference/nnmnkwii/).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers on `device`, row-major, densely
 *    packed.  Pointers suffixed _h are HOST pointers (tiny window tables).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *    only enqueue work; they never synchronise.  `status`/`path_len` are filled
 *    on the stream: read them after synchronising it.
 *  - dtype codes: MLPG_HIP_F32 = 0, MLPG_HIP_F64 = 1.
 *  - Return value: 0 on success, negative MLPG_HIP_E* on argument/runtime
 *    errors (text via mlpg_hip_last_error()).  Numerical failures (a leading
 *    minor that is not positive definite) are reported per system in `status`,
 *    never as a return code.
 *  - Windows are the reference's `(l, u, coeff)` triples packed as win_l_h[nw],
 *    win_u_h[nw], win_coef_h[sum(l+u+1)] (float64), window 0 first
 *    (paramgen/_mlpg.py:13-50).  Feature columns are window-major: column
 *    w*static_dim + d (paramgen/_mlpg.py:187-188).
 *  - Batches are the reference's zero-padded (B, Tmax, D) arrays
 *    (util/__init__.py:44-66, datasets/__init__.py:152-218); `lengths` (device
 *    int32[B], or NULL for "all Tmax") gives the valid frames per utterance.
 *    Output frames >= lengths[b] are zero-filled.
 *  - Re-entrant per (device, stream); internal scratch is cached per device
 *    and released by mlpg_hip_shutdown().  No OpenMP, no global mutable state
 *    besides that cache and the last-error string (thread-local).
 */
#ifndef MLPG_HIP_H_
#define MLPG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLPG_HIP_F32 0
#define MLPG_HIP_F64 1

#define MLPG_HIP_EINVAL (-1)   /* bad argument (shape, dtype, window table)   */
#define MLPG_HIP_ERUNTIME (-2) /* HIP runtime error (launch, malloc, device)  */
#define MLPG_HIP_ENOMEM (-3)   /* scratch allocation failed                   */

/* variance modes */
#define MLPG_HIP_VAR_FRAME 0   /* var is (B, Tmax, D)                         */
#define MLPG_HIP_VAR_GLOBAL 1  /* var is (D,), tiled over frames (_mlpg.py:169-170) */
#define MLPG_HIP_VAR_UNIT 2    /* var == NULL: unit variances (_mlpg.py:297-373) */

/* kernel selection for the forward/backward solves */
#define MLPG_HIP_ALGO_AUTO 0    /* strip kernel for wide streams, wave-per-system for narrow, else generic */
#define MLPG_HIP_ALGO_GENERIC 1 /* thread-per-system, factor in HBM scratch       */
#define MLPG_HIP_ALGO_WAVE 2    /* wave-per-system, factor in registers           */
#define MLPG_HIP_ALGO_STRIP 3   /* lane-per-static-dim, wavefront per 16-frame chunk, any T.  A stream of 1 .. 32 static dims (forward,
                                   three windows): the lanes of a wavefront run over
                                   64 / dims consecutive utterances x the dims instead (the transposed form, round 5) */
#define MLPG_HIP_ALGO_PIPE 4    /* retired in ABI 11 (the software-pipelined strip kernel of round 3; its sources left the tree in
                                   round 6 and are in the git history under tools/experimental/pipe): selects the strip kernel */
#define MLPG_HIP_ALGO_CONST 5   /* global (D,) / unit variances: the matrix of a static dim is the same for every
                                   utterance (_mlpg.py:169-170 tiles the variances) -- factorised once per launch, the
                                   solves are constant-coefficient recurrences, lane-per-static-dim, any T */
#define MLPG_HIP_ALGO_CHUNK 6   /* window extents up to 2 (5-tap windows: P has half-bandwidth 4), forward and backward: chunks of 16 + 4 frames
                                   eliminated twice around a block-tridiagonal solve over their separators; no workgroup waits
                                   for another; lane-per-static-dim, any T */
#define MLPG_HIP_ALGO_FIR 7     /* unit variances on float32 tensors (autograd.unit_variance_mlpg, autograd/_impl/mlpg.py:108-172: the
                                   reference multiplies by a dense float32 R): P^-1 is Toeplitz away from the ends and decays by
                                   about a bit per frame -- a 49-tap FIR filter plus 24 table rows per end, every 32-frame tile
                                   independent; the decay is checked per window set, else MLPG_HIP_EINVAL (AUTO: another kernel) */

int mlpg_hip_abi_version(void);
const char *mlpg_hip_last_error(void);
/* Test aid, not part of the reference's interface: launches per MLPG kernel family since the library was loaded --
 * 0 natural-order, 1 wave-per-system, 2 strip, 3 strip with several streams merged, 4 constant-coefficient, 5 fused, 6 chunked, 7 FIR,
 * 8 constant-coefficient with several streams merged, 9 strip in its transposed form (narrow streams: the lanes over several
 * utterances); and, counting CALLS rather than launches, 10 host-memory calls that took the short path with their inputs copied to the
 * device, 11 with the kernel reading the pinned staging buffer itself (see mlpg_hip_forward_host); 100 + d: chunks the chunked
 * host-memory calls (mlpg_hip_forward_host_multi / mlpg_hip_fastdtw_host_multi) have enqueued on device d; -1 for any other `kind`.
 * (Tests use it to assert WHICH kernel / route a call took.) */
long long mlpg_hip_launch_count(int kind);
int mlpg_hip_device_count(void);
/* Frees the per-device scratch caches. */
void mlpg_hip_shutdown(void);

/*
 * One training step of unit-variance MLPG under a mean-squared-error loss, fused: what the reference's
 * perf/autograd_mlpg_perf.py:56-86 loop does per batch with autograd.unit_variance_mlpg (autograd/_impl/mlpg.py:70-172:
 * dense R @ means forward, R^T @ grad backward) around torch.nn.MSELoss --
 *     y = MLPG(mean) with unit variances;  loss = sum_{live frames} (y - target)^2 / n_elems;
 *     grad_mean = d loss / d mean
 * in ONE kernel launch (wave-per-system scheme: both solves of a system share the launch, the trajectory stays in
 * registers between them).  mean (B, Tmax, D), target (B, Tmax, D/nw), grad_mean (B, Tmax, D) of `dtype`; y_out
 * (B, Tmax, D/nw) or NULL; loss: one float64 on the device; n_elems: the divisor of the mean (B * Tmax * D/nw for
 * nn.MSELoss over a padded batch).  Window extents <= 1, Tmax <= 1024.  The loss is summed in a fixed order
 * (bitwise repeatable).  status as for mlpg_hip_forward (may be NULL).
 * workspace: caller-owned device memory, 128-byte aligned, >= mlpg_hip_unit_mse_workspace_bytes(B, D, nw) bytes,
 * zeroed ONCE by the caller before its first use (the kernel leaves its arrival counter zero); one workspace per stream
 * that may run the call concurrently.  The call allocates nothing and is capturable into a HIP graph.
 * Float32 batches without lengths, Tmax >= 96, 1-3 windows of extent <= 2 the first of which is a single tap, given a workspace of
 * mlpg_hip_unit_mse_workspace_bytes_t(B, Tmax, D, nw) bytes: the step runs in the FIR form (MLPG_HIP_ALGO_FIR) as TWO launches --
 * forward + loss terms + d loss / d y into the workspace, then backward + the loss sum (fixed order, bitwise repeatable) -- with no
 * limit on Tmax; the first call for a window set builds the tap table (synchronous, not capturable: warm up before capturing).
 * With the smaller workspace of mlpg_hip_unit_mse_workspace_bytes the call behaves as before.
 */
int mlpg_hip_unit_mse_step(int device, void *stream, int dtype, const void *mean, const void *target,
                           const int32_t *lengths, int B, int Tmax, int D, int num_windows,
                           const int32_t *win_l_h, const int32_t *win_u_h, const double *win_coef_h,
                           double n_elems, void *y_out, void *grad_mean, double *loss, int32_t *status,
                           void *workspace, size_t workspace_bytes);
size_t mlpg_hip_unit_mse_workspace_bytes(int B, int D, int num_windows);
size_t mlpg_hip_unit_mse_workspace_bytes_t(int B, int Tmax, int D, int num_windows);
/*
 * Which form mlpg_hip_unit_mse_step takes for this problem when it is given the workspace of mlpg_hip_unit_mse_workspace_bytes_t:
 * 2 the FIR form, 1 the one-launch kernel, 0 neither (the step would return MLPG_HIP_EINVAL: the caller runs forward, loss and
 * backward as separate calls); negative: an error code.  The step decides with the same function.  The FIR form's conditions include
 * a property of the window set's NUMBERS (its inverse must decay to 2^-26 within 24 frames -- false for, e.g., dynamic windows scaled
 * by 4 or a static weight of 0.3), so the shape alone does not tell; the first query for a window set on a device builds the tap
 * table (synchronous).  While `stream` is being captured a table that does not exist yet cannot be built: the answer is then 1 or
 * 0, and may become 2 later (ABI 13).
 */
int mlpg_hip_unit_mse_form(int device, void *stream, int dtype, int has_lengths, int B, int Tmax, int D, int num_windows,
                           const int32_t *win_l_h, const int32_t *win_u_h, const double *win_coef_h);

/*
 * Measurement aid, not part of the reference's interface: a plain streaming copy of nbytes (a multiple of 16; both
 * pointers 16-byte aligned device memory), 16 bytes per lane and access.  bench.py times it in the same run as the
 * MLPG kernels to report the HBM rate a copy kernel reaches on the box ("fraction of both nominal and
 * measured-copy peak").
 */
int mlpg_hip_stream_copy(int device, void *stream, const void *src, void *dst, size_t nbytes);

/*
 * MLPG forward, batched.  Replaces paramgen.mlpg (paramgen/_mlpg.py:92-199)
 * and everything beneath it: build_win_mats (:13-50), build_poe (:53-89),
 * _bandmat/tensor.pyx dot_mv_plus_equals (:20-64) / dot_mm_plus_equals
 * (:82-174) and _bandmat/linalg.pyx solveh (:290-304) = _cholesky_banded
 * (:36-104) + _solve_triangular_banded (:106-176) x2.
 *
 *   out[b, :, d] = (sum_w W_w^T diag(tau_w) W_w)^-1 sum_w W_w^T (tau_w * mu_w)
 *   tau_w = 1/var evaluated in the INPUT dtype (:188), zeroed on the first/last
 *   max(max(l,u)) frames for w >= 1 (:177,191-193); all later arithmetic float64.
 *
 *   mean : (B, Tmax, D) dtype      var : per var_mode      out : (B, Tmax, D/nw) dtype
 *   status : int32 (B * D/nw): 0, or k = "k-th leading minor not positive
 *            definite" (linalg.pyx:79-82); may be NULL.
 */
int mlpg_hip_forward(int device, void *stream, int dtype, int algo,
                     const void *mean, const void *var, int var_mode,
                     const int32_t *lengths, int B, int Tmax, int D,
                     int num_windows, const int32_t *win_l_h,
                     const int32_t *win_u_h, const double *win_coef_h,
                     void *out, int32_t *status);

/*
 * The same call on HOST memory (numpy in, numpy out; no framework tensor in between): what a user of the
 * reference writes as a Python loop of paramgen.mlpg over a padded batch (util/__init__.py:44-66).  Synchronous.
 * The batch is cut into utterance chunks that alternate between two internal HIP streams, so that the host ->
 * device copy of one chunk, the kernels of another and the device -> host copy of a third overlap; pageable
 * memory is staged through pinned buffers by a few copy threads, memory from mlpg_hip_host_alloc (or any pinned /
 * registered host memory) is transferred in place.  All pointers are HOST pointers; shapes, dtypes, windows,
 * var_mode and status as for mlpg_hip_forward (lengths_h may be NULL).
 * A SMALL call -- at most 64 MB of input (MLPG_HIP_HOST_SMALL_MB) on one device: the literal per-utterance
 * paramgen.mlpg(mean_frames (T, D), variance_frames, windows) of the reference (_mlpg.py:92), 9.6 KB at T = 100 x 2 static
 * dims, 2.9 MB at T = 1000 x 60, and batches of a few utterances -- takes a short path instead: one stream, one cached pinned
 * staging buffer, no chunk plan and no thread; the arrays are staged and sent one behind the other (up to 48 KB the kernel reads
 * the pinned buffer itself), the kernel writes trajectory and verdicts straight into pinned host memory, and the host polls a
 * sequence number written behind them (round 6; mlpg_hip_launch_count(10 / 11) counts these calls).  An array of at least 1.2 MB
 * (MLPG_HIP_HOST_DIRECT_KB) that the library has been handed before (same address and size, one of its last 16), and any array of
 * at least 4 MB (MLPG_HIP_HOST_DIRECT_ALWAYS_KB), is not staged: the runtime copies it straight from -- the result: to -- the
 * caller's pageable memory at the pinned rate (mapping pages the device has never seen costs more than staging them, up to a few MB).
 * While the device works the calling thread touches the result array's pages (a store of 0 per page: the array must not overlap
 * the inputs), so that a fresh array's page faults are not taken one by one in the copy out (MLPG_HIP_HOST_PREFAULT=0: off).
 */
int mlpg_hip_forward_host(int device, int dtype, int algo, const void *mean_h,
                          const void *var_h, int var_mode,
                          const int32_t *lengths_h, int B, int Tmax, int D,
                          int num_windows, const int32_t *win_l_h,
                          const int32_t *win_u_h, const double *win_coef_h,
                          void *out_h, int32_t *status_h);

/*
 * mlpg_hip_fastdtw for N utterance pairs held in HOST memory: what DTWAligner.transform does per pair
 * (preprocessing/alignment.py:46-50: trim_zeros_frames on both utterances, fastdtw) for the whole batch, cut into
 * chunks of pairs that alternate between two internal streams (pinned staging, transfers under the kernels of the
 * other chunk), like mlpg_hip_forward_host.  X_h (N, Tx, D), Y_h (N, Ty, D) of `dtype` (float32 is widened to
 * float64 on the device: the distances are float64 either way).  lenx_h / leny_h: valid frames per utterance, or
 * both NULL: trailing frames with sum_d |x| < trim_eps are dropped on the device (preprocessing/generic.py:291-332,
 * eps 1e-7 there).  Outputs as mlpg_hip_fastdtw, in host memory; lenx_out_h / leny_out_h (may be NULL) receive the
 * lengths that were used.  Blocking; one host-entry call at a time per process.
 */
int mlpg_hip_fastdtw_host(int device, int dtype, const void *X_h, const void *Y_h,
                          const int32_t *lenx_h, const int32_t *leny_h, int N, int Tx, int Ty, int D,
                          int radius, int dist_kind, double dist_scale, int tie_rule, double trim_eps,
                          int32_t *path_i_h, int32_t *path_j_h, int32_t *path_len_h, double *cost_h,
                          int32_t *lenx_out_h, int32_t *leny_out_h);
/*
 * The two host-memory calls over SEVERAL devices from one process (the reference's batch loops --
 * util/__init__.py:44-66 over utterances, preprocessing/alignment.py:45 over pairs -- have no dependence between
 * items, so the batch shards with no exchange).  devices[0 .. num_devices): HIP device indices; NULL / 0: every visible
 * device.  Chunk c of the batch runs on list entry c % n, on that entry's stream pair slot (c / n) % 2; per device the
 * behaviour is the single-device call's, results and verdicts land at the chunk's own offset of the caller's arrays.
 * A device may be listed up to 4 times (each occurrence has its own staging buffers and streams).  On an error every
 * stream that was used is drained before the call returns.  mlpg_hip_forward_host(d, ...) == _multi(&d, 1, ...).
 * Environment (read once): MLPG_HIP_HOST_CHUNK_MB (input bytes per chunk, default 64), MLPG_HIP_HOST_COPY_THREADS (threads of the
 * staging copy, default 16, at most 64), MLPG_HIP_HOST_TRACE=1 (one line per call on stderr: where the calling thread and the
 * collector thread spent their time).
 */
int mlpg_hip_forward_host_multi(const int32_t *devices, int num_devices, int dtype, int algo, const void *mean_h,
                                const void *var_h, int var_mode, const int32_t *lengths_h, int B, int Tmax, int D,
                                int num_windows, const int32_t *win_l_h, const int32_t *win_u_h,
                                const double *win_coef_h, void *out_h, int32_t *status_h);
int mlpg_hip_fastdtw_host_multi(const int32_t *devices, int num_devices, int dtype, const void *X_h, const void *Y_h,
                                const int32_t *lenx_h, const int32_t *leny_h, int N, int Tx, int Ty, int D,
                                int radius, int dist_kind, double dist_scale, int tie_rule, double trim_eps,
                                int32_t *path_i_h, int32_t *path_j_h, int32_t *path_len_h, double *cost_h,
                                int32_t *lenx_out_h, int32_t *leny_out_h);
/*
 * The chunk dealing of the host-memory calls, as a pure function (no device needed): n_items utterances / pairs in
 * chunks of min(target_items, ceil(n_items / (4 * num_devices))) items (at least 4 chunks per device when the batch
 * allows).  Fills, for the first max_chunks chunks, the device-list entry, the stream slot, the first item and the
 * item count (any array may be NULL); returns the number of chunks, negative on bad arguments.
 */
long long mlpg_hip_host_chunk_plan(long long n_items, long long target_items, int num_devices, long long max_chunks,
                                   int32_t *entry, int32_t *slot, int64_t *first, int64_t *count);

/* Pinned host memory for arrays that are handed to mlpg_hip_forward_host repeatedly (transferred in place). */
void *mlpg_hip_host_alloc(size_t bytes);
void mlpg_hip_host_free(void *p);
/* Test aid, no GPU involved: dst[0, bytes) = src[0, bytes) by the short path's staging copy -- the calling thread and a few helper
 * threads (MLPG_HIP_HOST_HELPERS, default 3) share 64 KB slices claimed through one atomic word; the helpers spin for
 * MLPG_HIP_HOST_SPIN_US (default 250) after a copy, then sleep.  mlpg_hip_shutdown joins them. */
int mlpg_hip_host_copy(void *dst, const void *src, size_t bytes);

/*
 * Multi-stream MLPG forward over one padded acoustic feature batch (SURVEY 8(f)
 * rank 3).  Replaces the per-utterance, per-stream Python loop that users of
 * the reference write around paramgen.mlpg: util.apply_each2d_padded /
 * apply_each2d_trim (util/__init__.py:19-66) applied to each stream's column
 * slice of the (N, Tmax, D) zero-padded arrays that
 * datasets.PaddedFileSourceDataset.asarray produces (datasets/__init__.py:
 * 152-218), with the stream layout of util/files.py:90-115 (mgc | lf0 | vuv |
 * bap, each stream window-major: static, delta, delta-delta).
 *
 *   mean : (B, Tmax, ld_in) dtype; stream k occupies columns
 *          [in_col, in_col + max(num_windows,1)*static_dim)
 *   var  : MLPG_HIP_VAR_FRAME: (B, Tmax, ld_in), same columns;
 *          MLPG_HIP_VAR_GLOBAL: (ld_in,); MLPG_HIP_VAR_UNIT: NULL
 *   out  : (B, Tmax, ld_out) dtype; stream k's trajectory goes to columns
 *          [out_col, out_col + static_dim); other columns are not touched
 *   windows of stream k: entries win_first .. win_first+num_windows-1 of the
 *          packed tables (total_windows entries; several streams may share
 *          them).  num_windows == 0: no dynamic features, the static columns
 *          are copied through (frames >= lengths[b] zero-filled).
 *   status : int32 (B, sum_k static_dim), streams in table order; may be NULL.
 * The slices are consumed in place (no repacking).  Streams that share their
 * three windows (extents <= 1) are solved by ONE launch -- per-frame variances:
 * of the strip kernel; global (ld_in,) or unit variances (round 5): of the
 * constant-coefficient kernel -- their static dims side by side on the lanes; a
 * stream may be cut to fill the last 64-lane group, its remaining dims then run
 * as a launch of their own; every other stream is one launch.  A narrow stream (or
 * such a remainder) takes the strip kernel's
 * transposed form (see MLPG_HIP_ALGO_STRIP) behind the merged launch on `stream`.  Launches other than the widest go
 * to internal streams forked from and joined back into `stream` with events
 * (no host synchronisation; capturable): when the call returns, everything is
 * ordered on `stream`.  Which kernel solves a dim depends on the grouping, the
 * results agree to rounding (every kernel is held to the same parity bar).
 * Limits: 1 <= num_streams <= 64 (MLPG_HIP_EINVAL beyond), at most 4 streams
 * share one merged launch.
 */
typedef struct {
  int32_t in_col;      /* first column of the stream in mean / var rows  */
  int32_t out_col;     /* first column of its trajectory in out rows     */
  int32_t static_dim;  /* static feature dimension of the stream         */
  int32_t num_windows; /* 0 = pass-through                               */
  int32_t win_first;   /* first window of the stream in the window tables */
} mlpg_hip_stream_t;

int mlpg_hip_forward_streams(int device, void *stream, int dtype, int algo,
                             const void *mean, const void *var, int var_mode,
                             int64_t ld_in, const int32_t *lengths, int B,
                             int Tmax, int num_streams,
                             const mlpg_hip_stream_t *streams_h,
                             int total_windows, const int32_t *win_l_h,
                             const int32_t *win_u_h, const double *win_coef_h,
                             void *out, int64_t ld_out, int32_t *status);

/*
 * MLPG backward (gradient w.r.t. the means), batched.  Replaces
 * paramgen.mlpg_grad (paramgen/_mlpg.py:202-281; the reference solves a dense
 * T x T right-hand side with LAPACK dgbsv per static dim and window, :275).
 * Here: z = P_d^-1 o_d with the same banded factor as the forward, then
 *   grad[b, t, w*sd+d] = tau_w[t] * sum_k coeff_w[l_w+k] * z[t+k]      (O(T)).
 *
 *   var      : as in mlpg_hip_forward (in_dtype); grad_out : (B, Tmax, sd) in_dtype
 *   grad_mean: (B, Tmax, D) out_dtype (the reference returns float32, :248)
 * With var_mode == MLPG_HIP_VAR_UNIT this is also the backward of
 * autograd.UnitVarianceMLPG (autograd/_impl/mlpg.py:145-172: R^T . grad).
 */
int mlpg_hip_backward(int device, void *stream, int in_dtype, int out_dtype,
                      int algo, const void *var, int var_mode,
                      const void *grad_out, const int32_t *lengths, int B,
                      int Tmax, int D, int num_windows,
                      const int32_t *win_l_h, const int32_t *win_u_h,
                      const double *win_coef_h, void *grad_mean,
                      int32_t *status);

/*
 * The same call on HOST memory (ABI 14): the literal paramgen.mlpg_grad(mean_frames, variance_frames, windows, grad_output) of the
 * reference (paramgen/_mlpg.py:202-281: numpy in, float32 numpy out; 0.8 ms there at T = 100 x 2 static dims, seconds at T = 1000 x
 * 60) and the backward of autograd.MLPG on CPU tensors (autograd/_impl/mlpg.py:57-67).  All pointers are HOST pointers: var_h /
 * grad_out_h of in_dtype, grad_h (B, Tmax, D) of out_dtype, status_h (B * D/nw, may be NULL), lengths_h may be NULL.  Synchronous.
 * The batch runs through the short path of mlpg_hip_forward_host -- one stream, cached pinned staging, up to 48 KB of input read
 * by the kernel in place, the gradient written by the kernel into pinned host memory, a polled sequence number -- in pieces of
 * whole utterances of at most MLPG_HIP_HOST_SMALL_MB of input, one after the other (the reference's call is one utterance; a large
 * batch belongs on mlpg_hip_backward).  mlpg_hip_launch_count(10 / 11) counts the pieces.
 */
int mlpg_hip_backward_host(int device, int in_dtype, int out_dtype, int algo,
                           const void *var_h, int var_mode, const void *grad_out_h,
                           const int32_t *lengths_h, int B, int Tmax, int D,
                           int num_windows, const int32_t *win_l_h,
                           const int32_t *win_u_h, const double *win_coef_h,
                           void *grad_h, int32_t *status_h);

/*
 * Delta features (the step BEFORE MLPG in every pipeline).  Replaces
 * preprocessing.delta_features (preprocessing/generic.py:229-288: a Python loop over feature
 * dims of np.correlate(x[:, d], window, "same")), batched:
 *   out[b, t, w*D + d] = sum_{k=-l_w}^{u_w} coeff_w[l_w + k] * x[b, t + k, d]      (x = 0 outside [0, len_b))
 * x : (B, Tmax, D) dtype, out : (B, Tmax, D*nw) dtype; accumulation in float64.  For a window
 * array of length L the reference's "same" correlation is l = L-1-(L-1)/2, u = (L-1)/2.
 */
int mlpg_hip_delta_features(int device, void *stream, int dtype, const void *x,
                            const int32_t *lengths, int B, int Tmax, int D,
                            int num_windows, const int32_t *win_l_h,
                            const int32_t *win_u_h, const double *win_coef_h,
                            void *out);

/*
 * Modulation spectrum of parameter trajectories (the step
 * after MLPG).  Replace preprocessing/modspec.py: modspec (:6-53: power of the
 * rfft along time), inv_modspec (:62-100: irfft of sqrt(ms) * phase),
 * modspec_smoothing (:103-167: remove the modulation-frequency bins >= limit_bin
 * -- in the log domain the removed log-power is set to 0, i.e. unit magnitude
 * with the original phase -- and transform back), and the gradient of the
 * power spectrum of autograd/_impl/modspec.py:30-60 (a Python loop over feature
 * dimensions with dense (n/2+1, T) cos/sin tables).
 * Any DFT length n >= 2, as numpy's rfft / irfft.  A power of two <= 4096 (the
 * reference's defaults are 2048 and 4096): one workgroup per (utterance, pair of
 * feature columns), the n-point FFT lives in LDS, one launch per call.  Any
 * other n: the direct transform (O(n) per output value, exact integer phases;
 * two launches for smoothing / backward, the half spectrum in stream scratch).
 * All arrays float64, row-major; `ortho` selects numpy's norm="ortho".
 *   x     : (B, T, D), T <= n, zero-padded to n internally
 *   ms    : (B, n/2+1, D)            phase : (B, n/2+1, D, 2) (re, im), may be NULL
 *   inv_modspec output (B, n, D); smoothing / backward output (B, T, D)
 *   backward: grad_x[t] = C * sum_k grad_ms[k] (Re S_k cos(2 pi k t/n) - Im S_k sin(2 pi k t/n)),
 *             C = 2 (2/sqrt(n) for ortho), as modspec.py's analytic gradient.
 */
int mlpg_hip_modspec(int device, void *stream, const double *x, int B, int T,
                     int D, int n, int ortho, double *ms, double *phase);
int mlpg_hip_inv_modspec(int device, void *stream, const double *ms,
                         const double *phase, int B, int n, int D, int ortho,
                         double *x);
int mlpg_hip_modspec_smoothing(int device, void *stream, const double *x,
                               int B, int T, int D, int n, int ortho,
                               int limit_bin, int log_domain, double *out);
int mlpg_hip_modspec_backward(int device, void *stream, const double *x,
                              const double *grad_ms, int B, int T, int D,
                              int n, int ortho, double *grad_x);
/* Testing aid: on != 0 routes every DFT length through the direct transform
 * (process-wide), so that tests can compare it with the FFT path. */
void mlpg_hip_modspec_set_direct(int on);

/*
 * Trailing-zero trim.  Replaces preprocessing.trim_zeros_frames with trim="b"
 * (preprocessing/generic.py:291-332) applied to every utterance of a padded
 * (N, T, D) batch: lengths[n] = number of frames left after dropping trailing
 * frames whose sum_d |x| < eps.
 */
int mlpg_hip_trim_lengths(int device, void *stream, int dtype, const void *X,
                          int N, int T, int D, double eps, int32_t *lengths);

/*
 * fastdtw(x, y, radius, dist=L2) for N utterance pairs.  Replaces the
 * per-pair call in DTWAligner.transform (preprocessing/alignment.py:48-54;
 * fastdtw itself is the third-party package slaypni/fastdtw, see
 * oracle/dtw_oracle.c for the restated semantics and tie rule).
 *
 *   X : (N, Tx, D) float64, Y : (N, Ty, D) float64, lenx/leny : int32[N] valid
 *   frames (>= 1).  path_i/path_j : int32 (N, Tx+Ty) 0-based index pairs,
 *   path_len : int32[N] (0 if the pair failed), cost : float64[N] = accumulated
 *   distance D[len_x, len_y] (alignment.py:50, before the :51 normalisation).
 */
/* local distances of mlpg_hip_fastdtw */
#define MLPG_HIP_DIST_L2 0           /* sqrt(sum_k (x_k-y_k)^2), k ascending: DTWAligner's default dist
                                        (alignment.py:35: lambda x, y: norm(x - y)) in the oracle's summation order */
#define MLPG_HIP_DIST_SCALED_L2_NP 1 /* dist_scale * sqrt(((x-y)*(x-y)).sum()) with the sum in numpy's pairwise
                                        order (D <= 128): metrics.melcd(x, y) for two frames
                                        (metrics/__init__.py:27-59) with dist_scale = 10/ln(10)*sqrt(2) */
#define MLPG_HIP_DIST_SCALED_L1_NP 2 /* dist_scale * np.abs(x-y).sum(): city-block distance, numpy's pairwise order */
#define MLPG_HIP_DIST_SCALED_SQL2_NP 3 /* dist_scale * ((x-y)**2).sum(): squared Euclidean, numpy's pairwise order
                                        (what DTWAligner(dist=...) callables of those forms resolve to; the reference
                                        accepts any Python callable, alignment.py:35 -- the kernel evaluates the cost) */
/* tie rules of the DP recurrence D[i,j] = dt + min(D[i-1,j], D[i,j-1], D[i-1,j-1]) when candidates are EQUAL (after
 * adding dt).  The reference does not pin fastdtw's version, and the package ships two implementations of the
 * recurrence that differ only here; continuous data never tie, quantised or shifted-copy data
 * (tests/test_preprocessing.py:441-456 aligns zero-shifted copies) do. */
#define MLPG_HIP_TIE_FIRST_MIN 0 /* upstream's pure-Python __dtw: min(..., key=) keeps the FIRST minimum of
                                    (up, left, diagonal) -- the rule every parity test of this repo is pinned on */
#define MLPG_HIP_TIE_DIAG_LAST 1 /* the strict-less chain recalled for upstream's compiled _fastdtw: up only if it
                                    beats both others, else left only if it beats the diagonal, else the diagonal.
                                    UNVERIFIED: the package is absent from this image (and from /root/reference);
                                    tests/test_dtw_gpu.py reports which rule an installed fastdtw follows */
int mlpg_hip_fastdtw(int device, void *stream, const double *X,
                     const double *Y, const int32_t *lenx,
                     const int32_t *leny, int N, int Tx, int Ty, int D,
                     int radius, int dist_kind, double dist_scale,
                     int tie_rule, int32_t *path_i, int32_t *path_j,
                     int32_t *path_len, double *cost);
/*
 * fastdtw for `dist` callables no device-side cost reproduces (preprocessing/alignment.py:35-38 accepts ANY Python
 * callable): one resolution level at a time, the local costs evaluated on the host by the user's callable, the DP
 * recurrence, the back-trace and the window expansion (upstream's __expand_window in interval form) on the device.
 * The caller walks the levels from the coarsest to the finest (nnmnkwii_amd/preprocessing/alignment.py shows how):
 *
 *   mlpg_hip_dtw_level_windows   level_tx / level_ty: int32[N], the pair's lengths at THIS level (0: the pair's
 *       recursion has not reached the level yet, it is skipped); full[n] != 0: the pair's recursion bottoms out here
 *       (len < radius + 2: full window); otherwise the window comes from cpath_* = the pair's path at the level below
 *       (coarser), as mlpg_hip_dtw_level_from_costs left it.  Out: row_lo / row_hi (N, row_stride) inclusive column
 *       intervals, row_off (N, row_stride + 1) prefix sums of the widths (row_off[n, level_tx[n]] = cells of pair n).
 *   mlpg_hip_dtw_level_from_costs   costs: float64, pair n's cells at costs[cost_base[n] + row_off[n, i] + j - row_lo[n, i]]
 *       (total_cells doubles in all), max_ty >= every level_ty.  Out: path_i / path_j (N, path_stride), path_len[n]
 *       (0: the corner is unreachable), cost[n] = D[len_x, len_y].  tie_rule: MLPG_HIP_TIE_*.
 * All pointers are device pointers; both calls only enqueue work on `stream`.
 */
int mlpg_hip_dtw_level_windows(int device, void *stream, int N, int radius,
                               const int32_t *level_tx, const int32_t *level_ty,
                               const int32_t *full, const int32_t *cpath_i,
                               const int32_t *cpath_j, const int32_t *cpath_len,
                               int cpath_stride, int32_t *row_lo, int32_t *row_hi,
                               int64_t *row_off, int row_stride);
int mlpg_hip_dtw_level_from_costs(int device, void *stream, int N, int tie_rule,
                                  const int32_t *level_tx, const int32_t *level_ty,
                                  const int32_t *row_lo, const int32_t *row_hi,
                                  const int64_t *row_off, int row_stride, int max_ty,
                                  const double *costs, const int64_t *cost_base,
                                  int64_t total_cells, int32_t *path_i,
                                  int32_t *path_j, int32_t *path_len,
                                  int path_stride, double *cost);
/* = mlpg_hip_fastdtw(..., MLPG_HIP_DIST_L2, 1.0, MLPG_HIP_TIE_FIRST_MIN, ...) */
int mlpg_hip_fastdtw_l2(int device, void *stream, const double *X,
                        const double *Y, const int32_t *lenx,
                        const int32_t *leny, int N, int Tx, int Ty, int D,
                        int radius, int32_t *path_i, int32_t *path_j,
                        int32_t *path_len, double *cost);

/*
 * Frame-wise conversion under a joint source/target GMM.  Replaces the per-frame,
 * per-mixture Python loop of baseline/gmm.py:97-120 (MLPGBase.transform: np.linalg.solve(covarXX[m], x - mu_x[m])
 * for every frame and mixture) and :225-244 (MLPG.transform: the same for the most likely mixture of each frame):
 *   out[n, :] = sum_m posterior[n, m] * (mu_y[m] + A[m] (x[n] - mu_x[m])),     A[m] = covarYX[m] covarXX[m]^-1
 * x (N, D), posterior (N, M) or NULL, mixture int32 (N) (used when posterior is NULL: one mixture per frame),
 * mu_x (M, D), mu_y (M, Dy), A (M, Dy, D), out (N, Dy); all float64, row-major.  A is factored once per model by
 * the caller (host LAPACK); the mixture posteriors come from scikit-learn as in the reference.
 */
int mlpg_hip_gmm_convert(int device, void *stream, const double *x,
                         const double *posterior, const int32_t *mixture,
                         const double *mu_x, const double *mu_y,
                         const double *A, int64_t N, int D, int Dy, int M,
                         double *out);

/*
 * Gather rows along the warping path into zero-padded outputs.  Replaces
 * alignment.py:52-54,72-73:  out[n, k, :] = src[n, path[n, k], :] for
 * k < path_len[n], zeros after.  src (N, Tsrc, D), out (N, Tout, D), same dtype.
 */
int mlpg_hip_gather_path(int device, void *stream, int dtype, const void *src,
                         const int32_t *path, const int32_t *path_len, int N,
                         int Tsrc, int path_stride, int D, int Tout, void *out);

#ifdef __cplusplus
}
#endif
#endif /* MLPG_HIP_H_ */
