// Layout conversion at the C-ABI boundary and small utility kernels.
#include "dc_device.h"

namespace dc {

// ---------------------------------------------------------------------------------------------------
// layout conversion at the boundary: host float64 xyz-interleaved  <->  device float32 planar
// ---------------------------------------------------------------------------------------------------
__global__ void k_f64i_to_f32p(const double *__restrict__ src, float *__restrict__ dst, int n, long total, const int *__restrict__ user_of) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const double *s = src + (b * n + (user_of ? user_of[i] : i)) * 3;     // device vertex i <- caller's vertex user_of[i]
  float *d = dst + b * 3 * n;
  d[i] = (float) s[0]; d[n + i] = (float) s[1]; d[2 * n + i] = (float) s[2];
}
__global__ void k_f64i_to_f64p(const double *__restrict__ src, double *__restrict__ dst, int n, long total, const int *__restrict__ user_of) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const double *s = src + (b * n + (user_of ? user_of[i] : i)) * 3;
  double *d = dst + b * 3 * n;
  d[i] = s[0]; d[n + i] = s[1]; d[2 * n + i] = s[2];
}
__global__ void k_f32p_to_f64i(const float *__restrict__ src, double *__restrict__ dst, int n, long total, const int *__restrict__ user_of) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const float *s = src + b * 3 * n;
  double *d = dst + (b * n + (user_of ? user_of[i] : i)) * 3;
  d[0] = s[i]; d[1] = s[n + i]; d[2] = s[2 * n + i];
}
// the same conversions for buffers that already live on the device in the caller's layout (torch tensors: fp32 or fp64, xyz
// interleaved) — the device-pointer boundary dc_*_dev
template <class T>
__global__ void k_i_to_f32p(const T *__restrict__ src, float *__restrict__ dst, int n, long total, const int *__restrict__ user_of) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const T *s = src + (b * n + (user_of ? user_of[i] : i)) * 3;
  float *d = dst + b * 3 * n;
  d[i] = (float) s[0]; d[n + i] = (float) s[1]; d[2 * n + i] = (float) s[2];
}
template <class T>
__global__ void k_f32p_to_i(const float *__restrict__ src, T *__restrict__ dst, int n, long total, const int *__restrict__ user_of) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const float *s = src + b * 3 * n;
  T *d = dst + (b * n + (user_of ? user_of[i] : i)) * 3;
  d[0] = (T) s[i]; d[1] = (T) s[n + i]; d[2] = (T) s[2 * n + i];
}
template <class T>
__global__ void k_copy_cast(const float *__restrict__ src, T *__restrict__ dst, long total) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t < total) dst[t] = (T) src[t];
}

__global__ void k_seed_gradient(const float *__restrict__ x, const float *__restrict__ target, float *__restrict__ gx,
                                float *__restrict__ gv, int n3, long total, float scale) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int k = (int) (t % n3);
  gx[t] = scale * (x[t] - target[k]);
  gv[t] = 0.f;
}

void launch_f64i_to_f32p(const double *src, float *dst, int B, int n, const int *user_of, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  hipLaunchKernelGGL(k_f64i_to_f32p, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, dst, n, total, user_of);
}
void launch_f64i_to_f64p(const double *src, double *dst, int B, int n, const int *user_of, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  hipLaunchKernelGGL(k_f64i_to_f64p, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, dst, n, total, user_of);
}
void launch_f32p_to_f64i(const float *src, double *dst, int B, int n, const int *user_of, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  hipLaunchKernelGGL(k_f32p_to_f64i, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, dst, n, total, user_of);
}
void launch_dev_to_planar(const void *src, int is_f32, float *dst, int B, int n, const int *user_of, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  if (is_f32) hipLaunchKernelGGL(k_i_to_f32p<float>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, (const float *) src, dst, n, total, user_of);
  else hipLaunchKernelGGL(k_i_to_f32p<double>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, (const double *) src, dst, n, total, user_of);
}
void launch_planar_to_dev(const float *src, void *dst, int is_f32, int B, int n, const int *user_of, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  if (is_f32) hipLaunchKernelGGL(k_f32p_to_i<float>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, (float *) dst, n, total, user_of);
  else hipLaunchKernelGGL(k_f32p_to_i<double>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, (double *) dst, n, total, user_of);
}
void launch_copy_cast(const float *src, void *dst, int is_f32, long total, hipStream_t st) {
  if (total == 0) return;
  if (is_f32) hipLaunchKernelGGL(k_copy_cast<float>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, (float *) dst, total);
  else hipLaunchKernelGGL(k_copy_cast<double>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, (double *) dst, total);
}
void launch_seed_gradient(const float *x, const float *target, float *gx, float *gv, int B, int N, float scale, hipStream_t st) {
  long total = (long) B * 3 * N;
  hipLaunchKernelGGL(k_seed_gradient, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, x, target, gx, gv, 3 * N, total, scale);
}

}  // namespace dc
