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
  const int n = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)Tout * D) return;
  const int k = (int)(e / D), c = (int)(e % D);
  T v = (T)0;
  if (k < path_len[n] && k < path_stride) {
    const int r = path[(size_t)n * path_stride + k];
    if (r >= 0 && r < Tsrc) v = src[((size_t)n * Tsrc + r) * D + c];
  }
  out[((size_t)n * Tout + k) * D + c] = v;
}

}  // namespace

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
  dim3 grid((unsigned)((per + 255) / 256), (unsigned)N);
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
