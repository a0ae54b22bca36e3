// Developer tool: calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this part for the access widths the step kernels
// use (4 B per lane, coalesced: 256 B per wave), as MI355X_MICROARCH.md (HBM section) asks before trusting absolutes.
// Each kernel moves a known byte count over a 1 GiB buffer (4 x the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read4(const float *a, size_t n, float *out) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456f) out[0] = s;
}
__global__ void k_read16(const float4 *a, size_t n4, float *out) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) { float4 q = a[i]; s += q.x + q.y + q.z + q.w; }
  if (s == 123.456f) out[0] = s;
}
__global__ void k_write4(float *a, size_t n) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) a[i] = (float) i;
}
__global__ void k_write16(float4 *a, size_t n4) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) a[i] = make_float4(1, 2, 3, 4);
}
int main() {
  const size_t n = (size_t) 1 << 28;     // 1 GiB of floats
  float *a, *o;
  hipMalloc(&a, n * 4); hipMalloc(&o, 4);
  hipMemset(a, 0, n * 4);
  for (int rep = 0; rep < 2; rep++) {
    k_read4<<<2048, 256>>>(a, n, o);
    k_read16<<<2048, 256>>>((const float4 *) a, n / 4, o);
    k_write4<<<2048, 256>>>(a, n);
    k_write16<<<2048, 256>>>((float4 *) a, n / 4);
  }
  hipDeviceSynchronize();
  std::printf("each kernel moves %zu bytes\n", n * 4);
  return 0;
}
