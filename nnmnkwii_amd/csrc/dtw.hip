// DTW-side kernels: trailing-zero trim, path gather, fastdtw (L2).
#include "common.h"

namespace mlpg {
namespace {

// lengths[n] = frames left after dropping trailing frames with sum_d |x| < eps
// (preprocessing/generic.py:291-332, trim="b").  One wave per utterance: lanes
// stride over frames from the end, the first (highest) non-zero frame wins.
template <typename T>
__global__ __launch_bounds__(64) void trim_kernel(const T *__restrict__ X, int Tn, int D, double eps,
                                                  int32_t *__restrict__ lengths) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const T *x = X + (size_t)n * Tn * D;
  int last = -1;  // highest frame index whose |x| sum is >= eps
  for (int base = Tn - 1; base >= 0 && last < 0; base -= 64) {
    const int t = base - lane;
    bool nz = false;
    if (t >= 0) {
      T s = (T)0;  // summed in the input dtype like np.sum(np.abs(x), axis=1)
      for (int k = 0; k < D; ++k) s += (x[(size_t)t * D + k] < (T)0 ? -x[(size_t)t * D + k] : x[(size_t)t * D + k]);
      nz = !((double)s < eps);
    }
    const unsigned long long m = __ballot(nz);
    if (m) last = base - (__ffsll((long long)m) - 1);
  }
  if (lane == 0) lengths[n] = last + 1;
}

// out[n, k, :] = src[n, path[n, k], :] for k < path_len[n], zero after.
template <typename T>
__global__ void gather_kernel(const T *__restrict__ src, const int32_t *__restrict__ path,
                              const int32_t *__restrict__ path_len, int Tsrc, int path_stride, int D, int Tout,
                              T *__restrict__ out) {
  // the utterance index is folded into blockIdx.x (gridDim.y is limited to 65535): nbx blocks per utterance
  const int nbx = (int)(((long)Tout * D + blockDim.x - 1) / blockDim.x);
  const int n = blockIdx.x / nbx;
  const long e = (long)(blockIdx.x % nbx) * blockDim.x + threadIdx.x;
  if (e >= (long)Tout * D) return;
  const int k = (int)(e / D), c = (int)(e % D);
  T v = (T)0;
  if (k < path_len[n] && k < path_stride) {
    const int r = path[(size_t)n * path_stride + k];
    if (r >= 0 && r < Tsrc) v = src[((size_t)n * Tsrc + r) * D + c];
  }
  out[((size_t)n * Tout + k) * D + c] = v;
}

// Frame-wise GMM conversion E[y | x_n] = sum_m post[n, m] (mu_y[m] + A[m] (x_n - mu_x[m])), A[m] = S_yx[m] S_xx[m]^-1
// (the per-frame, per-mixture np.linalg.solve loop of baseline/gmm.py:97-120, 225-244; A is factored once per model on
// the host).  One thread per (frame, output dim); post = NULL with mix != NULL selects one mixture per frame.
__global__ void gmm_convert_kernel(const double *__restrict__ x, const double *__restrict__ post,
                                   const int32_t *__restrict__ mix, const double *__restrict__ mu_x,
                                   const double *__restrict__ mu_y, const double *__restrict__ A, long N, int D, int Dy,
                                   int M, double *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * Dy) return;
  const long n = e / Dy;
  const int d = (int)(e % Dy);
  const double *xn = x + n * D;
  double acc = 0.0;
  const int m0 = mix ? mix[n] : 0, m1 = mix ? m0 + 1 : M;
  for (int m = m0; m < m1; ++m) {
    const double *a = A + ((size_t)m * Dy + d) * D;
    const double *mx = mu_x + (size_t)m * D;
    double v = mu_y[(size_t)m * Dy + d];
    for (int k = 0; k < D; ++k) v += a[k] * (xn[k] - mx[k]);
    acc += (post ? post[n * M + m] : 1.0) * v;
  }
  out[e] = acc;
}

}  // namespace

int launch_gmm_convert(hipStream_t s, const double *x, const double *post, const int32_t *mix, const double *mu_x,
                       const double *mu_y, const double *A, long N, int D, int Dy, int M, double *out) {
  const long total = N * Dy;
  if (total == 0) return 0;
  hipLaunchKernelGGL(gmm_convert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, post, mix, mu_x, mu_y, A,
                     N, D, Dy, M, out);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_trim(hipStream_t s, int dtype, const void *X, int N, int T, int D, double eps, int32_t *lengths) {
  if (dtype == MLPG_HIP_F32)
    hipLaunchKernelGGL(trim_kernel<float>, dim3(N), dim3(64), 0, s, (const float *)X, T, D, eps, lengths);
  else
    hipLaunchKernelGGL(trim_kernel<double>, dim3(N), dim3(64), 0, s, (const double *)X, T, D, eps, lengths);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_gather(hipStream_t s, int dtype, const void *src, const int32_t *path, const int32_t *path_len, int N,
                  int Tsrc, int path_stride, int D, int Tout, void *out) {
  const long per = (long)Tout * D;
  const long nblk = ((per + 255) / 256) * (long)N;
  if (nblk > 0x7fffffffL) {
    set_error("gather_path: problem too large (%ld blocks)", nblk);
    return MLPG_HIP_EINVAL;
  }
  dim3 grid((unsigned)nblk);
  if (dtype == MLPG_HIP_F32)
    hipLaunchKernelGGL(gather_kernel<float>, grid, dim3(256), 0, s, (const float *)src, path, path_len, Tsrc,
                       path_stride, D, Tout, (float *)out);
  else
    hipLaunchKernelGGL(gather_kernel<double>, grid, dim3(256), 0, s, (const double *)src, path, path_len, Tsrc,
                       path_stride, D, Tout, (double *)out);
  MLPG_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mlpg
